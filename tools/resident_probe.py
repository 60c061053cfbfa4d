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
    for n_val_patches in (0, 16, 512):
        times = []
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
            st.patch(plist)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            root = st.hash_tree_root()
            times.append(time.perf_counter() - t0)
        ref = ssz.hash_tree_root_beacon_state_deneb(bytes(enc), 0)
        print(f"{n_val_patches:4d} validator records + 8192 balance/flag patches per slot: resident root "
              f"{1e3 * min(times):.3f} ms (median {1e3 * sorted(times)[len(times) // 2]:.3f}), equals the from-scratch root: {root == ref}",
              flush=True)
    import ctypes
    from ethereum_consensus_amd import _lib
    L = _lib.load()
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
