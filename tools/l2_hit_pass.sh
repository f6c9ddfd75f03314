# L2 hit / miss counts per kernel of one Klein-9B fp8 forward pair (PMC pass; kernels serialised).  usage: tools/l2_hit_pass.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/l2hit
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/l2hit -o l2 -- python3 $R/tools/one_forward.py klein-9b 2 fp8 > $R/gpurun_out/l2hit.log 2>&1 || { tail -5 $R/gpurun_out/l2hit.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
f=glob.glob(f'{R}/gpurun_out/l2hit/**/*counter_collection.csv', recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'gemm256s' not in n and 'attn64b_kernel' not in n: continue
    key=n.split('(')[0][-60:]
    acc[key][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='TCC_HIT_sum': cnt[key]+=1
for k,v in acc.items():
    h,m=v.get('TCC_HIT_sum',0),v.get('TCC_MISS_sum',0)
    print(f"{k:62s} launches {cnt[k]:4d}  hit {h/cnt[k]:.3e}  miss {m/cnt[k]:.3e}  hit rate {h/(h+m+1e-9):.3f}  req {v.get('TCC_REQ_sum',0)/cnt[k]:.3e}")
PY
