#!/bin/bash
# Run ON the GPU box (through gpurun):  bash tools/profile_round.sh <tag> [extra bench flags]
# Collects, for the default bench workload (C3), the rocprofv3 kernel-trace statistics and -- in SEPARATE passes, with --kernel-trace
# only, as MI355X_MICROARCH.md prescribes -- the FETCH_SIZE / WRITE_SIZE counters, and reduces the result databases (too large to hand
# back) to small summaries under gpurun_out/: <tag>_kernel_stats.txt, <tag>_pmc_hbm_traffic.json.  Copy those into profiles/.
set -u
TAG=${1:-rXX}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary $*"
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $CMD > $OUT/${TAG}_kt_bench.json 2> /tmp/kt.err
DB=$(find /tmp/p_kt -name "*.db" | head -1)
python $REPO/tools/prof_summarize.py stats "$DB" $OUT/${TAG}_kernel_stats.txt "rocprofv3 --kernel-trace --stats -- $CMD ($TAG; warm-up update + 1 timed update + acting pass of the synthetic rollout)" > /dev/null
if [ "${SKIP_PMC:-0}" != "1" ]; then
  CMD0="python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-secondary $*"
  SVLA_GEMM_LOG=/tmp/gemm_f.log rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- $CMD0 > /dev/null 2> /tmp/f.err
  SVLA_GEMM_LOG=/tmp/gemm_w.log rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- $CMD0 > /dev/null 2> /tmp/w.err
  FDB=$(find /tmp/p_f -name "*.db" | head -1); WDB=$(find /tmp/p_w -name "*.db" | head -1)
  python $REPO/tools/prof_summarize.py pmc "$FDB" "$WDB" $OUT/${TAG}_pmc_hbm_traffic.json > /dev/null
  cmp -s /tmp/gemm_f.log /tmp/gemm_w.log || echo "WARNING: the two passes launched different GEMM sequences"
  python $REPO/tools/prof_summarize.py pmc_shapes "$FDB" "$WDB" /tmp/gemm_f.log $OUT/${TAG}_pmc_hbm_traffic_by_shape.json
fi
if [ "${SKIP_PREPROC:-0}" != "1" ]; then
  # observation-tensor loads (north_star: "coalesced HBM loads of the (T.B, C, H, W) observation tensor evidenced by rocprof HBM-GB/s"): the u8 frame kernels
  # (normalize_u8, patchify_u8) and the rest of the frozen ViT, same separate-pass recipe
  VCMD="python $REPO/tools/vit_probe.py"
  rm -rf /tmp/v_kt /tmp/v_f /tmp/v_w
  rocprofv3 --kernel-trace --stats -d /tmp/v_kt -o kt -- $VCMD > /dev/null 2> /tmp/vkt.err
  python $REPO/tools/prof_summarize.py stats "$(find /tmp/v_kt -name '*.db' | head -1)" $OUT/${TAG}_vit_kernel_stats.txt "frozen DINOv2 ViT-S/14 probe + normalize_u8 (tools/vit_probe.py; $TAG)" > /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/v_f -o f -- $VCMD > /dev/null 2> /tmp/vf.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/v_w -o w -- $VCMD > /dev/null 2> /tmp/vw.err
  python $REPO/tools/prof_summarize.py pmc "$(find /tmp/v_f -name '*.db' | head -1)" "$(find /tmp/v_w -name '*.db' | head -1)" $OUT/${TAG}_pmc_preproc_traffic.json > /dev/null
fi
ls -la $OUT | grep "$TAG"
