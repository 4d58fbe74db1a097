#!/bin/bash
# L2 <-> fabric traffic of the shipped library from rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md
# prescribes): one 512-block compression batch (-d3 -q2, one scheduler instance) and one 2400-block decoding pass.
# Writes profiles/r06_pmc_final.txt (per kernel) and profiles/r06_pmc_final.json (what bench.py reads; it names the sources it is valid for).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
D=gpurun_out/pmc_tmp; rm -rf $D; mkdir -p $D
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $D/cf -- python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 1 --warmup 0 --decode-blocks 0 > /dev/null 2> $D/cf.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $D/cw -- python bench.py --no-cpu --pipeline 1 --blocks 512 --steps 1 --warmup 0 --decode-blocks 0 > /dev/null 2> $D/cw.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $D/df -- python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 1 > /dev/null 2> $D/df.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $D/dw -- python tools/decode_bench.py --blocks 2400 --distinct 300 -d 3 -q 2 --passes 1 > /dev/null 2> $D/dw.err
python - $D <<'PY'
import glob, json, os, sqlite3, subprocess, sys
from collections import defaultdict
sys.path.insert(0, os.getcwd())
import bench
D = sys.argv[1]

def counters(sub, name):
    agg = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(D, sub, "**", "*.db"), recursive=True):
        db = sqlite3.connect(path)
        for kname, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
            if cname == name:
                a = agg[kname.replace(".kd", "")]; a[0] += 1; a[1] += val
    return agg

def short(k):
    k = k.split("(")[0]
    return k[5:] if k.startswith("void ") else k

out = {"csrc_sha": bench.csrc_sha(), "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "working tree",
       "unit": "FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them; bytes = (FETCH x 2 + WRITE) x 1024 for the compression kernels (16 B/lane streaming reads are tallied at half on gfx950), (FETCH + WRITE) x 1024 for the decoding kernels (64-byte row accesses)"}
lines = []
for what, fsub, wsub, blocks in (("compress", "cf", "cw", 512), ("decode", "df", "dw", 2400)):
    f, w = counters(fsub, "FETCH_SIZE"), counters(wsub, "WRITE_SIZE")
    ker = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] * 2 + w.get(k, [0, 0])[1])):
        if k.startswith("k_synth") or "synth" in k:
            continue
        ker[short(k)] = {"launches": max(f.get(k, [0, 0])[0], w.get(k, [0, 0])[0]), "fetch_kib": f.get(k, [0, 0])[1], "write_kib": w.get(k, [0, 0])[1]}
    mult = 2 if what == "compress" else 1
    tot = lambda pred: sum((v["fetch_kib"] * mult + v["write_kib"]) * 1024 for k, v in ker.items() if pred(k)) / blocks
    sec = {"blocks": blocks, "kernels": ker}
    if what == "compress":
        is_dec = lambda k: k.startswith("k_dec") or k.startswith("k_selftest") or k.startswith("k_lds_order")
        sec["all_bytes_per_block"] = tot(lambda k: not is_dec(k))
        sec["k_rc_bytes_per_block"] = tot(lambda k: k == "k_rc" or k.startswith("k_rcs"))      # the range coder: k_rcs<16/32> (k_rc: the redo list, normally empty)
        sec["k_part_bytes_per_block"] = tot(lambda k: k == "k_part")
        sec["model_bytes_per_block"] = tot(lambda k: k.startswith("k_model") or k in ("k_binoff", "k_place"))
        sec["text_bytes_per_block"] = tot(lambda k: k in ("k_count_lines", "k_scan_tiles", "k_index_lines", "k_records", "k_prep_stats", "k_prep_write", "k_rec_offsets") or k.startswith("k_tag"))
    else:
        sec["bytes_per_block"] = tot(lambda k: k.startswith("k_dec"))
    out[what] = sec
    lines.append(f"== {what}: {blocks} blocks, bytes per block = (FETCH x {mult} + WRITE) x 1024 / blocks")
    for k, v in ker.items():
        lines.append(f"{k[:56]:56s} n={v['launches']:5d} FETCH {v['fetch_kib']:14.0f} KiB  WRITE {v['write_kib']:14.0f} KiB  -> {(v['fetch_kib'] * mult + v['write_kib']) * 1024 / blocks / 1e6:9.2f} MB per block")
    lines.append("   ".join(f"{k} = {v / 1e6:.1f} MB" for k, v in sec.items() if k.endswith("per_block")))
os.makedirs("profiles", exist_ok=True)
json.dump(out, open("profiles/r06_pmc_final.json", "w"), indent=1)
open("profiles/r06_pmc_final.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(l for l in lines if l.startswith("==") or "per_block" in l or " = " in l))
PY
mkdir -p gpurun_out/profiles && cp profiles/r06_pmc_final.json profiles/r06_pmc_final.txt gpurun_out/profiles/
rm -rf $D
