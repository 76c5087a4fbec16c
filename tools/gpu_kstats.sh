#!/bin/bash
# rocprofv3 kernel-trace stats of one E. coli-shape zmo run of bin/wtzmo built with the given extra flags: per-kernel totals
# usage: tools/gpu_kstats.sh "<hipcc flags>" [dmo]
export TMPDIR=/tmp; R=$(pwd)
python - <<'PY' > /dev/null 2>&1
import sys; sys.path.insert(0,'.')
import bench
bench.gen_reads('/tmp/ecoli.fa',4600000,25.0,11)
PY
WTZ_HIPCC_FLAGS="$1" python -c "import __graft_entry__ as g; g.build_product(force=True)" > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
if [ "$2" = dmo ]; then A="-k 16 -z 10 -Z 16 -U -1 -m 0.1 -A 1000"; else A="-k 16 -s 200 -m 0.6"; fi
cd /tmp && rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- $R/bin/wtzmo --pool-gb 60 -i /tmp/ecoli.fa -fo /tmp/v.ovl $A > /tmp/ks.log 2>&1
grep -E "records,|kernel ms" /tmp/ks.log; md5sum /tmp/v.ovl | cut -c1-32
python - <<'PY'
import csv,re,glob
f=glob.glob('/tmp/ks/**/ks_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    n=r['Name']; m=re.search(r'<(K_\w+)',n); n=m.group(1) if m else n[:40]
    print(f"{n:28s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:10.1f} max_us {float(r['MaxNs'])/1e3:10.1f}")
PY
