#!/bin/bash
# A/B of the tracked frame on ONE box: tools/dev/frame_ab.sh <libA> <libB> [reps]  — the moving-camera sequence (tools/dev/track_seq.py)
# with the two libraries alternating, `reps` times each; prints every run and the two medians (box-to-box spread is +-2 us:
# variants are only comparable inside one call)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
A=$1; B=$2; N=${3:-5}
for i in $(seq $N); do
  for v in A B; do
    lib=$([ $v = A ] && echo $A || echo $B)
    us=$(PTAM_HIP_LIB=$lib python $R/tools/dev/track_seq.py 6 2>/dev/null | grep -o "[0-9.]* us/frame" | cut -d' ' -f1)
    echo "$v $us"
  done
done | tee /tmp/frame_ab.txt
python3 - <<PY
import statistics
a = [float(l.split()[1]) for l in open("/tmp/frame_ab.txt") if l.startswith("A")]
b = [float(l.split()[1]) for l in open("/tmp/frame_ab.txt") if l.startswith("B")]
print("A median %.1f (min %.1f)  B median %.1f (min %.1f)  B - A = %.1f us" % (statistics.median(a), min(a), statistics.median(b), min(b), statistics.median(b) - statistics.median(a)))
PY
