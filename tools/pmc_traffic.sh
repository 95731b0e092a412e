#!/bin/bash
# HBM traffic of the fmx kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, as
# MI355X_MICROARCH.md prescribes), on the GPU box:  tools/pmc_traffic.sh <tag> [bench args...]
# Writes gpurun_out/<tag>_front_pmc.json (copy it to profiles/ to have bench.py report `roofline.traffic`).
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -o t -- python $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python - "$TAG" "$@" <<'PY'
import sqlite3, json, sys, glob
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob("gpurun_out/pmc_%s/t_results.db" % c)[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? and kernel_name like 'fmx::%' or kernel_name like 'void fmx::%' group by kernel_name", (c,))
    out[c + "_KB_per_launch"] = {r[0].split("(")[0]: r[1] for r in rows}
line = [l for l in open("gpurun_out/pmc_FETCH_SIZE.log") if l.startswith("{")]
cfg = json.loads(line[-1])["config"] if line else {}
ch, n = cfg.get("channels_per_gpu"), cfg.get("block_samples_per_channel")
import re
isfront = lambda k: re.search(r"front\d?_kernel", k) is not None and "pre" not in k       # front_kernel / f3::front3_kernel / f4::front4_kernel: whichever ran
f = sum(v for k, v in out["FETCH_SIZE_KB_per_launch"].items() if isfront(k))
w = sum(v for k, v in out["WRITE_SIZE_KB_per_launch"].items() if isfront(k))
res = {
    "note": "rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs) of `python bench.py %s` on MI355X; per launch of each fmx kernel. "
            "FETCH_SIZE is doubled for the front kernel per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced streaming reads); units are KB." % " ".join(sys.argv[2:]),
    "workload": cfg.get("workload"), "channels": ch, "block": n,
    **out,
    "front_kernel_hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024),
    "front_kernel_algorithmic_bytes_per_launch": int(8.0 * ch * n + 8.0 * ch * n / 12) if ch else None,
}
json.dump(res, open("gpurun_out/%s_front_pmc.json" % tag, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("channels", "block", "front_kernel_hbm_bytes_per_launch", "front_kernel_algorithmic_bytes_per_launch")}))
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
