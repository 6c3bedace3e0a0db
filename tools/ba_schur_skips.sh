# timing experiments on the two-part Schur kernel (results invalid): 16 no block loop, 64 no g, 128 no reduction, 256 no products
for sk in ${SKIPS:-0 16 64 80 128 256 384}; do
  echo "skip=$sk"; ORBSLAM_AMD_BA_SCHUR_SKIP=$sk ORBSLAM_AMD_BA_SPLIT=1 ORBSLAM_AMD_BA_PHASE_PROF=1 timeout -k 10 120 python tools/ba_probe.py 32 2>&1 | grep -E "phase kernels" | tail -1
done
