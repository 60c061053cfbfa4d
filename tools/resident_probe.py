"""Development probe: per-slot state root of a device-resident deneb state (2^20 validators) after a slot's worth of
patches -- full re-Merkleization from the records vs the resident state's cached validator roots."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from ethereum_consensus_amd import ssz, synthetic  # noqa: E402


def main(n=1 << 20):
    r = random.Random(3)
    f = synthetic.state_fields(n, "mainnet", seed=5)
    enc = bytearray(synthetic.serialize_state(f))
    st = ssz.ResidentBeaconStateDeneb(bytes(enc), 0)
    fixed = len(enc) - sum(len(x) for x in (
        f["historical_roots"].tobytes(), b"x" * 72 * len(f["eth1_data_votes"]), f["validators"].tobytes(), f["balances"].tobytes(),
        f["previous_epoch_participation"].tobytes(), f["current_epoch_participation"].tobytes(), f["inactivity_scores"].tobytes(),
        synthetic.serialize_payload_header(f["payload_header"]), f["historical_summaries"].tobytes()))
    vals_off = fixed + len(f["historical_roots"].tobytes()) + 72 * len(f["eth1_data_votes"])
    bal_off = vals_off + 121 * n
    part_off = bal_off + 8 * n + n  # current_epoch_participation
    t0 = time.perf_counter()
    root0 = st.hash_tree_root()
    t1 = time.perf_counter()
    print(f"first root (builds the validator-root cache): {1e3 * (t1 - t0):.2f} ms", flush=True)
    import ctypes
    from ethereum_consensus_amd import _lib
    L = _lib.load()
    L.ecgpu_prof_enable(1)
    L.ecgpu_prof_filter(None)

    def prof(tag):
        ms, n = ctypes.c_double(), ctypes.c_uint64()
        L.ecgpu_prof_read(tag.encode(), ctypes.byref(ms), ctypes.byref(n))
        return ms.value, n.value
    for n_val_patches in (0, 16, 512):
        times, ptimes, hashes = [], [], []
        p0 = {t: prof(t) for t in ("merkle_tree_climb", "merkle_state_tail", "merkle_tree_rebuild")}
        for slot in range(12):
            patches = {}
            for _ in range(4096):  # ~2^12 balances + participation flags per slot
                i = r.randrange(n)
                patches[bal_off + 8 * i] = r.randbytes(8)
                patches[part_off + r.randrange(n)] = bytes([r.randrange(8)])
            for _ in range(n_val_patches):  # effective balance / exit epoch of a few records
                patches[vals_off + 121 * r.randrange(n) + 80] = r.randbytes(8)
            plist = sorted(patches.items())
            for off, b in plist:
                enc[off:off + len(b)] = b
            offs = (ctypes.c_uint64 * len(plist))(*[o for o, _ in plist])
            doff = [0]
            for _, b in plist:
                doff.append(doff[-1] + len(b))
            doffs = (ctypes.c_uint64 * len(doff))(*doff)
            blob = b"".join(b for _, b in plist)
            tp = time.perf_counter()
            rc = L.ecgpu_resident_state_patch(st.handle, offs, doffs, blob, len(plist))  # the C entry alone (no Python marshalling)
            ptimes.append(time.perf_counter() - tp)
            assert rc == 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            root = st.hash_tree_root()
            times.append(time.perf_counter() - t0)
            hashes.append(int(L.ecgpu_last_hash64_count()))
        ref = ssz.hash_tree_root_beacon_state_deneb(bytes(enc), 0)
        print(f"{n_val_patches:4d} validator records + 8192 balance/flag patches per slot: resident root "
              f"{1e3 * min(times):.3f} ms (median {1e3 * sorted(times)[len(times) // 2]:.3f}), patch entry {1e3 * min(ptimes):.3f} ms "
              f"(median {1e3 * sorted(ptimes)[len(ptimes) // 2]:.3f}), hash64 per root {min(hashes)} .. {max(hashes)}, "
              f"equals the from-scratch root: {root == ref}", flush=True)
        if os.environ.get("ECGPU_TREE_TRACE"):
            tr = (ctypes.c_uint64 * (3 * 2048))()
            L.ecgpu_debug_tree_trace(tr)
            rows = [(tr[3 * i], tr[3 * i + 1], tr[3 * i + 2]) for i in range(2048) if tr[3 * i]]
            if rows:
                t0 = min(r_[0] for r_ in rows)
                print(f"      climb trace: {len(rows)} region workgroups; start spread {(max(r_[0] for r_ in rows) - t0) / 100:.1f} us, "
                      f"count pass avg {sum(r_[1] - r_[0] for r_ in rows) / len(rows) / 100:.1f} us, climb pass avg "
                      f"{sum(r_[2] - r_[1] for r_ in rows) / len(rows) / 100:.1f} us (max {max(r_[2] - r_[1] for r_ in rows) / 100:.1f}), "
                      f"last end {(max(r_[2] for r_ in rows) - t0) / 100:.1f} us", flush=True)
        for t in p0:
            ms, k = prof(t)
            if k > p0[t][1]:
                print(f"      {t}: {1e3 * (ms - p0[t][0]) / (k - p0[t][1]):.1f} us per launch ({k - p0[t][1]} launches)", flush=True)
    d = torch.frombuffer(bytearray(enc), dtype=torch.uint8).cuda()
    d_root = torch.empty(32, dtype=torch.uint8, device="cuda")
    fs = int(L.ecgpu_beacon_state_deneb_fixed_size(0))
    hfix = (ctypes.c_uint8 * fs).from_buffer_copy(bytes(enc[:fs]))
    s = torch.cuda.current_stream().cuda_stream
    ts = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.ecgpu_htr_beacon_state_deneb_dev(d.data_ptr(), len(enc), hfix, 0, d_root.data_ptr(), s)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"from the 121-byte records every time (device-resident encoding): {1e3 * min(ts):.3f} ms", flush=True)
    st.close()


if __name__ == "__main__":
    main()
