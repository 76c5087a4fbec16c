#!/bin/bash
# K_pair occupancy sweep on the per-engine kernels: -DWTZ_OCC_PAIR=4 / 5 / 6 (VGPR budget 128 / 102 / 85), configs[2] zmo + dmo kernel time
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
cp smartdenovo_amd/libwtzmo_hip.so /tmp/lib_keep.so
for occ in 4 6; do
  WTZ_HIPCC_FLAGS="-DWTZ_OCC_PAIR=$occ" python -c "import __graft_entry__ as g; g.build_product(force=True)" > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
  echo "== WTZ_OCC_PAIR=$occ"
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zmo', d['value'], d['ms_per_step'], d['kernel_ms_last_step']['pairs'], d['records_last_step'])"
  python bench.py --engine dmo --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dmo', d['value'], d['ms_per_step'], d['kernel_ms_last_step']['pairs'], d['records_last_step'])"
done
cp /tmp/lib_keep.so smartdenovo_amd/libwtzmo_hip.so
