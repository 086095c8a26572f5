#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 6144 9216 12288 18432 24576; do
  echo "== LIDARHIP_MC_EMIT_WAVES=$w"
  LIDARHIP_MC_EMIT_WAVES=$w bash tools/mc_kernels.sh 2>&1 | grep "emit_batch\|phase_ms" | cut -c1-150
done
