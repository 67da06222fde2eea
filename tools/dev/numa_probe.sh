#!/bin/bash
# which NUMA node is the GPU on, and how does the BA bench behave when the process is pinned to each node?
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) cpus=$(cat $d/local_cpulist 2>/dev/null) vendor=$(cat $d/vendor 2>/dev/null)"; done
lscpu | grep -i -E "numa|socket|model name|^CPU\(s\)" | head -12
for cpus in 0-15 64-79 128-143 192-207; do
  echo "== taskset -c $cpus"
  for i in 1 2 3; do taskset -c $cpus timeout 200 python bench.py --no-cpu-baseline --no-tracking 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step']*1e3,1))"; done
done
