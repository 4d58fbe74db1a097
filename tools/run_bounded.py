#!/usr/bin/env python3
"""Run a command in its own process group and kill the whole group after a time limit (GPU experiments must
never outlive their slot: a hung kernel otherwise holds the box until the outer limit).
Usage: tools/run_bounded.py <seconds> <command> [args...]"""
import os
import signal
import subprocess
import sys


def main():
    limit = float(sys.argv[1])
    p = subprocess.Popen(sys.argv[2:], start_new_session=True)
    try:
        sys.exit(p.wait(timeout=limit))
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        p.wait()
        print(f"[run_bounded] killed after {limit:.0f} s: {' '.join(sys.argv[2:])}", file=sys.stderr)
        sys.exit(124)


if __name__ == "__main__":
    main()
