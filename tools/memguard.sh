#!/bin/bash
# usage: memguard.sh <max_rss_GB> <cmd...> : run cmd, kill it if its process-group RSS exceeds the cap
# (keeps a runaway host allocation from taking the whole GPU box down).
cap_kb=$(( $1 * 1024 * 1024 )); shift
setsid "$@" &
pid=$!
while kill -0 $pid 2>/dev/null; do
  pg=$(ps -o pgid= -p $pid | tr -d ' ')
  rss=$(ps -o rss= -g "$pg" 2>/dev/null | awk '{s+=$1} END {print s+0}')
  if [ "${rss:-0}" -gt "$cap_kb" ]; then
    echo "[memguard] RSS ${rss} kB > cap ${cap_kb} kB: killing process group" >&2
    kill -9 -- -"$pg" 2>/dev/null
    break
  fi
  sleep 1
done
wait $pid
