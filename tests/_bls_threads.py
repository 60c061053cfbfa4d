"""BLS verification from MANY host threads ("no BLS concurrency parity test").  The reference's functions
are pure and re-entrant and its spec-test harness runs trials on a thread pool (spec-tests/main.rs:114-124, SURVEY.md 8b
"Threading"); here 16 threads share one process, one device and one validated-key registry.

    python -m tests._bls_threads <workload.pkl> <n_threads> <rounds>

Phase 1  every thread makes its FIRST call at the same instant (a barrier releases them; the main thread has not touched the
         library beyond loading it): ecgpu_init, the box self-check / tower decision, the row-program upload, stream sets and
         arenas are all created under contention.
Phase 2  each thread loops over batch sizes {1, 64, 700, 5 000} x {host keys, validated-key registry, collector flush} on a slice
         of the mutated corpus that starts at a thread- and round-specific offset; every status is compared with the C++ oracle's
         (tests/_bls_config2.prepare_mutated: judged on all tuples, the Python oracle on a sample).
The dispatch controls are read from the environment once per process, so the test runs this file once per setting
(ECGPU_FORK_THREADS_MAX default and 64).  Test infrastructure (the workload file holds oracle verdicts)."""
import json
import pickle
import sys
import threading
import time


def main(path: str, n_threads: int, rounds: int) -> int:
    from ethereum_consensus_amd import _lib, bls
    with open(path, "rb") as f:
        w = pickle.load(f)
    N, pks, msgs, sigs, want = w["n"], w["pks"], w["msgs"], w["sigs"], w["cpp"]
    L = _lib.load(build_if_missing=False)  # no ecgpu_* call yet: the threads' first calls race through initialisation
    errors, first_ms = [], [0.0] * n_threads
    start = threading.Barrier(n_threads)
    phase2 = threading.Barrier(n_threads + 1)
    go = threading.Barrier(n_threads + 1)
    reg_box = [None]
    counts = {"host": 0, "registry": 0, "collector": 0, "tuples": 0}
    lock = threading.Lock()

    def cut(lo, n):
        return pks[48 * lo:48 * (lo + n)], msgs[32 * lo:32 * (lo + n)], sigs[96 * lo:96 * (lo + n)], bytes(want[lo:lo + n])

    def worker(t):
        try:
            # phase 1: simultaneous first calls, a different shape per thread (a lone verify, small and mid batches)
            n1 = (1, 1, 64, 700, 1, 2000, 64, 1)[t % 8]
            p, m, s, exp = cut(97 * t, n1)
            start.wait()
            t0 = time.perf_counter()
            got = bls.fast_aggregate_verify_batch(p, None, m, s) if n1 > 1 else bytes([bls.verify_signature_status(p, m, s)])
            first_ms[t] = (time.perf_counter() - t0) * 1e3
            if got != exp:
                errors.append(("first call", t, n1, [i for i in range(n1) if got[i] != exp[i]][:4]))
            phase2.wait()
            go.wait()  # the main thread has filled the registry
            reg = reg_box[0]
            batch = bls.SignatureBatch(reg)
            for rnd in range(rounds):
                for k, n in enumerate((1, 64, 700, 5000)):
                    lo = (7919 * t + 104729 * rnd + 131 * k) % (N - n)
                    p, m, s, exp = cut(lo, n)
                    mode = ("host", "registry", "collector")[(t + rnd + k) % 3]
                    if mode == "host":
                        got = bls.fast_aggregate_verify_batch(p, None, m, s)
                    elif mode == "registry":
                        got = reg.fast_aggregate_verify_batch(list(range(lo, lo + n)), list(range(n + 1)), m, s)
                    else:
                        for i in range(n):
                            if i % 2:
                                batch.fast_aggregate_verify_indexed([lo + i], m[32 * i:32 * i + 32], s[96 * i:96 * i + 96])
                            else:
                                batch.verify_signature(p[48 * i:48 * i + 48], m[32 * i:32 * i + 32], s[96 * i:96 * i + 96])
                        got = batch.flush()
                    if got != exp:
                        errors.append((mode, t, rnd, n, lo, [(i, got[i], exp[i]) for i in range(n) if got[i] != exp[i]][:4]))
                    with lock:
                        counts[mode] += 1
                        counts["tuples"] += n
            batch.close()
        except Exception as e:  # noqa: BLE001
            errors.append(("exception", t, repr(e)))
            for b in (phase2, go):
                try:
                    b.abort()
                except Exception:  # noqa: BLE001
                    pass

    th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for x in th:
        x.start()
    try:
        phase2.wait(timeout=600)
        reg = bls.ValidatorKeyRegistry(N)
        reg.set(0, pks)
        reg_box[0] = reg
        go.wait(timeout=600)
    except threading.BrokenBarrierError:
        pass
    for x in th:
        x.join()
    ok = not errors and counts["host"] and counts["registry"] and counts["collector"]
    print(json.dumps({"ok": bool(ok), "threads": n_threads, "rounds": rounds, "calls": counts, "first_call_ms_max": round(max(first_ms), 1),
                      "first_call_ms_min": round(min(first_ms), 1), "tower": L.ecgpu_bls_tower(), "errors": [repr(e) for e in errors[:6]]}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3])))
