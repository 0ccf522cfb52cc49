# source me: run_limited SECONDS cmd...  — runs cmd in its own process group and kills the WHOLE group at the limit
# (rocprofv3 that aborts on an uncollectable counter set hangs in its signal handler together with the profiled process)
run_limited() {
  local lim=$1; shift
  setsid "$@" &
  local pid=$!
  ( exec > /dev/null 2>&1 < /dev/null; sleep $lim & s=$!; trap "kill $s; exit 0" TERM; wait $s; kill -KILL -- -$pid ) &
  local watchdog=$!
  wait $pid; local rc=$?
  kill $watchdog 2>/dev/null; wait $watchdog 2>/dev/null
  return $rc
}
