#!/bin/bash
# timing split of the band Cholesky by switching parts off (results invalid): tools/chol_skips.sh
# bits: 1 fill from S, 2 diagonal-tile factorisation, 4 backward substitution, 8 zeroing, 16 panel + trailing update
for nt in 512 256; do
  b=256; [ $nt = 256 ] && b=512
  for sk in 0 1 2 4 8 16 3 7 31; do
    echo -n "NT=$nt batch=$b skip=$sk: "
    ORBSLAM_AMD_BA_CHOL_NT=$nt ORBSLAM_AMD_BA_CHOL_SKIP=$sk BATCH=$b ORBSLAM_AMD_BA_PHASE_PROF= python3 $GRAFT_REPO_ROOT/tools/ba_probe.py 32 2>&1 | grep -o "Cholesky [0-9.]*" | tail -1
  done
done
