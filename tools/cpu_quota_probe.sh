#!/bin/bash
# Does the box throttle the process's CPU time (cgroup quota)?  prints the quota, and the throttle counters around a command
show() { for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us; do [ -r $f ] && echo "$f: $(cat $f)"; done
         for f in /sys/fs/cgroup/cpu.stat /sys/fs/cgroup/cpu/cpu.stat; do [ -r $f ] && { echo "$f:"; grep -E "throttled|nr_periods|usage_usec" $f | tr '\n' ' '; echo; }; done; }
echo "nproc $(nproc)  online $(cat /sys/devices/system/cpu/online)  affinity $(taskset -p $$ 2>/dev/null | awk '{print $NF}')"
cat /proc/self/cgroup | head -3
show
"$@"
show
