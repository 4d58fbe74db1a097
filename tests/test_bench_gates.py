"""bench.py's N > 1 step gating (CPU, no GPU, no torch): every scheduler instance has a gather thread of its own; the gather of
instance i's step s runs after i has finished step s, overlaps with i's step s+1 (and with the other instances' gathers), and i does
not start step s+2 (same output buffer) before it is done.  launch_ranks(): the ranks bench.py starts by itself get a launcher's
environment, and one failing rank takes the others with it."""
import importlib.util
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_step_gates_order():
    bench = _bench()
    P, first, total = 4, 1, 9
    gates = bench.StepGates(P)
    log = []; lock = threading.Lock()
    rng = random.Random(3)
    delays = {(i, s): rng.random() * 0.01 for i in range(P) for s in range(first, total)}

    def worker(i):
        for s in range(first, total):
            gates.lane_may_start(i, s, first)
            with lock:
                log.append(("start", i, s))
            time.sleep(delays[(i, s)])
            with lock:
                log.append(("done", i, s))
            gates.lane_done(i, s)

    def gatherer(i):
        for s in range(first, total):
            gates.wait_lane(i, s)
            with lock:
                log.append(("gather_begin", i, s))
            time.sleep(0.004)
            with lock:
                log.append(("gather_end", i, s))
            gates.lane_gathered(i, s)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(P)] + [threading.Thread(target=gatherer, args=(i,)) for i in range(P)]
    for t in th:
        t.start()
    for t in th:
        t.join(10)
        assert not t.is_alive()
    pos = {e: k for k, e in enumerate(log)}
    overlapped = False; side_by_side = False
    for s in range(first, total):
        for i in range(P):
            gb, ge = pos[("gather_begin", i, s)], pos[("gather_end", i, s)]
            assert pos[("done", i, s)] < gb                                   # an instance's gather only after it finished s
            if s + 2 < total:
                assert pos[("start", i, s + 2)] > ge                          # buffer of step s is not reused before its gather
            if s + 1 < total and pos[("start", i, s + 1)] < ge:
                overlapped = True
            for k in range(P):
                if k != i and pos[("gather_begin", k, s)] < ge and pos[("gather_end", k, s)] > gb:
                    side_by_side = True
    assert overlapped                                                          # the next step does run during a gather
    assert side_by_side                                                        # and the instances' gathers do not wait for each other


def test_launch_ranks_environment_and_failure(tmp_path, monkeypatch):
    bench = _bench()
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, sys, time\n"
        "r = int(os.environ['RANK'])\n"
        "open(os.path.join(os.path.dirname(__file__), 'env%d' % r), 'w').write(' '.join(os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))\n"
        "if os.environ.get('FAIL_RANK') == str(r): sys.exit(3)\n"
        "if os.environ.get('FAIL_RANK'): time.sleep(60)\n")
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
    assert bench.launch_ranks(3) == 0
    envs = [open(tmp_path / f"env{r}").read().split() for r in range(3)]
    assert [e[0] for e in envs] == ["0", "1", "2"] and [e[1] for e in envs] == ["0", "1", "2"]
    assert all(e[2] == "3" and e[3] == "127.0.0.1" for e in envs) and len({e[4] for e in envs}) == 1
    monkeypatch.setenv("DSRC_BENCH_SAME_GPU", "1")
    assert bench.launch_ranks(2) == 0
    assert [open(tmp_path / f"env{r}").read().split()[1] for r in range(2)] == ["0", "0"]
    monkeypatch.setenv("FAIL_RANK", "1")
    t0 = time.time()
    assert bench.launch_ranks(2) == 3
    assert time.time() - t0 < 30                                               # rank 0 (asleep for a minute) was not waited for
