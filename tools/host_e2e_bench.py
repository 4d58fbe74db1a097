#!/usr/bin/env python3
"""File -> archive throughput of the dsrc-amd CLI (C++ host pipeline over the C ABI, PCIe and file I/O included),
next to the reference CLI on the same file.  Input: the synthetic Illumina set written to tmpfs.
Usage: tools/host_e2e_bench.py [GB of input, default 6] [instances, default 4]"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from dsrc_amd._lib import Handle  # noqa: E402

CLI = os.path.join(ROOT, "dsrc_amd", "csrc", "dsrc-amd")
REF = os.path.join(ROOT, "oracle", "_ref", "dsrc_ref")


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    inst = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    src = os.path.join(d, "e2e.fastq"); ours = os.path.join(d, "e2e_ours.dsrc"); theirs = os.path.join(d, "e2e_ref.dsrc")
    h = Handle()
    recs_per_piece = 2_000_000
    total = 0; first = 1
    with open(src, "wb") as f:
        while total < gb * 1e9:
            cap = recs_per_piece * 400
            dptr = h.dev_alloc(cap)
            n = h.synth_illumina(first, recs_per_piece, dptr, cap)
            f.write(h.dev_download(dptr, n)); h.dev_free(dptr)
            total += n; first += recs_per_piece
    h.close()
    size = os.path.getsize(src)
    res = {}
    # extra switches, e.g. -n384; several variants separated by "/" (each is run twice): -n96 / -n96 -t6 / -n64 -t8
    variants = [v.split() for v in " ".join(sys.argv[3:]).split("/")] if len(sys.argv) > 3 else [[]]
    runs = []
    for extra in variants:
        name = "dsrc-amd -t%d %s" % (inst, " ".join(extra))
        cmd = [CLI, "c", "-d3", "-q2", f"-t{inst}", *extra, src, ours]
        runs += [(name, cmd), (name + " (2nd run)", cmd)]
    if not os.environ.get("E2E_NO_REF"):
        runs.append(("reference -t60", [REF, "c", "-d3", "-q2", "-t60", src, theirs]))
    for name, cmd in runs:
        if not os.path.exists(cmd[0]):
            continue
        if os.environ.get("E2E_GAP"):          # separate the runs: remove the previous archive, let the driver reclaim the previous process's memory
            if cmd[0] == CLI and os.path.exists(ours):
                os.remove(ours)
            time.sleep(float(os.environ["E2E_GAP"]))
        t = time.time(); subprocess.check_call(cmd); dt = time.time() - t
        res[name] = size / dt / 1e6
        print(f"{name:30s}: {size / 1e9:.2f} GB in {dt:6.2f} s = {size / dt / 1e6:8.1f} MB/s")
    if os.path.exists(theirs):
        same = md5(ours) == md5(theirs)
        print("archives identical:", same)
        assert same
    # the way back: archive -> FASTQ through the GPU decompressor, next to the reference's `dsrc d`
    back = os.path.join(d, "e2e_back.fastq")
    druns = [("dsrc-amd d -t%d" % inst, [CLI, "d", f"-t{inst}", ours, back])]
    if not os.environ.get("E2E_NO_REF") and os.path.exists(REF):
        druns.append(("reference d -t60", [REF, "d", "-t60", ours, back]))
    for name, cmd in druns:
        if os.path.exists(back):
            os.remove(back)
        t = time.time(); subprocess.check_call(cmd); dt = time.time() - t
        print(f"{name:30s}: {size / 1e9:.2f} GB in {dt:6.2f} s = {size / dt / 1e6:8.1f} MB/s")
        assert os.path.getsize(back) == size
        with open(src, "rb") as fa, open(back, "rb") as fb:          # spot comparison: 64 MiB pieces across the file
            for k in range(16):
                o = (size - (64 << 20)) * k // 15
                fa.seek(o); fb.seek(o)
                assert fa.read(64 << 20) == fb.read(64 << 20), f"decoded text differs near byte {o}"
    print("decoded text identical to the input (size + 16 x 64 MiB spot checks)")
    for p in (src, ours, theirs, back):
        if os.path.exists(p):
            os.remove(p)


if __name__ == "__main__":
    main()
