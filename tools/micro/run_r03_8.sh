mkdir -p gpurun_out/r03
export FDM_NO_REBUILD=1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace -d gpurun_out/r03/prof_w4 -o t -- python3 bench.py --steps 4 --warmup 2 > gpurun_out/r03/prof_w4.json 2> gpurun_out/r03/prof_w4.err
bash fastdm_amd/csrc/build.sh -DFDM_GEMM_W4=0 > gpurun_out/r03/build_old.log 2>&1
rocprofv3 --kernel-trace -d gpurun_out/r03/prof_old -o t -- python3 bench.py --steps 4 --warmup 2 > gpurun_out/r03/prof_old.json 2> gpurun_out/r03/prof_old.err
python3 - <<'PY'
import sqlite3, glob, collections
for tag in ("w4", "old"):
    db = glob.glob(f"gpurun_out/r03/prof_{tag}/**/*.db", recursive=True)
    if not db: print(tag, "no db"); continue
    con = sqlite3.connect(db[0]); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = list(cur.execute(f"select s.kernel_name, d.grid_size_x, d.workgroup_size_x, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id"))
    agg = collections.defaultdict(list)
    for n, g, w, dt in rows:
        if 'gemm' in n: agg[(n.split('(')[0][-40:], g // w)].append(dt)
    print("==", tag)
    for k in sorted(agg, key=lambda k: -sum(agg[k])):
        v = sorted(agg[k]); print(f"  {k[0]:42s} blocks {k[1]:5d} n {len(v):4d} median {v[len(v)//2]/1e3:8.1f} us total {sum(v)/1e6:8.2f} ms")
PY
rm -rf gpurun_out/r03/prof_w4 gpurun_out/r03/prof_old
