python tools/k7_only.py 50 5000 200 0 19 2>&1 | grep K7
# (the switches below exist in the measurement build only: make -C ptam_cg_amd/csrc ab)
export PTAM_HIP_LIB=${PTAM_HIP_LIB:-${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}/tools/_ab/libptam_hip.so}
for t in 256 512; do for w in 1 2 3 4 6 8; do
  echo -n "loop threads=$t wg/cu=$w: "; PTAM_K7_LOOP=1 PTAM_K7_THREADS=$t PTAM_K7_WG_PER_CU=$w python tools/k7_only.py 50 5000 200 0 19 2>&1 | grep K7
done; done
