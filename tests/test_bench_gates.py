"""bench.py's N > 1 step gating (CPU, no GPU, no torch): the gather of step s runs after every scheduler instance has
finished step s, overlaps with step s+1, and no instance starts step s+2 (same output buffer) before it is done."""
import importlib.util
import os
import random
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
            gates.lane_may_start(s, first)
            with lock:
                log.append(("start", i, s))
            time.sleep(delays[(i, s)])
            with lock:
                log.append(("done", i, s))
            gates.lane_done(s)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(P)]
    for t in th:
        t.start()
    for s in range(first, total):
        gates.wait_step(s)
        with lock:
            log.append(("gather_begin", -1, s))
        time.sleep(0.004)
        with lock:
            log.append(("gather_end", -1, s))
        gates.step_gathered(s)
    for t in th:
        t.join(10)
        assert not t.is_alive()
    pos = {e: k for k, e in enumerate(log)}
    overlapped = False
    for s in range(first, total):
        gb, ge = pos[("gather_begin", -1, s)], pos[("gather_end", -1, s)]
        for i in range(P):
            assert pos[("done", i, s)] < gb                                   # gather only after all instances finished s
            if s + 2 < total:
                assert pos[("start", i, s + 2)] > ge                          # buffer of step s is not reused before its gather
            if s + 1 < total and pos[("start", i, s + 1)] < ge:
                overlapped = True
    assert overlapped                                                          # the next step does run during a gather
