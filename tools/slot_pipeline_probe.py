"""Development probe for BASELINE configs[4] on ONE GPU: per slot, one eth_fast_aggregate_verify over the participating
keys of a 512-key sync committee (reference semantics: every key decompressed and checked; and through the validated-key
registry) and one state root of the device-resident 2^20-validator state after the slot's patches -- enqueued on two
streams, timed over 32 slots, next to the same work done one after the other."""
import hashlib
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ethereum_consensus_amd import _lib, bls, ssz, synthetic  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def S(tag, i):
    return hashlib.sha256(b"ecgpu/v1/" + tag + b"/" + i.to_bytes(4, "little")).digest()


def main(n=1 << 20, slots=32):
    L = _lib.load(build_if_missing=False)
    r = random.Random(11)
    dev = torch.device("cuda:0")
    sks = [1 + int.from_bytes(S(b"sync", i), "big") % (R - 1) for i in range(512)]
    pks = bls.sk_to_pk_batch(b"".join(s.to_bytes(32, "big") for s in sks))
    reg = bls.ValidatorKeyRegistry(512)
    reg.set(0, pks)
    f = synthetic.state_fields(n, "mainnet", seed=5)
    enc = bytearray(synthetic.serialize_state(f))
    st = ssz.ResidentBeaconStateDeneb(bytes(enc), 0)
    st.hash_tree_root()  # builds the validator-root cache
    tail = [f["balances"].tobytes(), f["previous_epoch_participation"].tobytes(), f["current_epoch_participation"].tobytes(),
            f["inactivity_scores"].tobytes(), synthetic.serialize_payload_header(f["payload_header"]), f["historical_summaries"].tobytes()]
    bal_off = len(enc) - sum(len(x) for x in tail)
    part_off = bal_off + 8 * n + n
    # per-slot inputs prepared up front (device-resident, as the metric wants)
    work = []
    for slot in range(slots):
        part = [i for i in range(512) if r.random() < 0.95]
        msg = S(b"slot", slot)
        sig = bls.sign_batch((sum(sks[i] for i in part) % R).to_bytes(32, "big"), [msg])
        keys = b"".join(pks[48 * i:48 * i + 48] for i in part)
        patches = {}
        for _ in range(4096):
            patches[bal_off + 8 * r.randrange(n)] = r.randbytes(8)
            patches[part_off + r.randrange(n)] = bytes([r.randrange(8)])
        work.append(dict(
            k=len(part), d_keys=torch.frombuffer(bytearray(keys), dtype=torch.uint8).to(dev),
            d_idx=torch.tensor(part, dtype=torch.int32, device=dev), d_off=torch.tensor([0, len(part)], dtype=torch.int32, device=dev),
            d_msg=torch.frombuffer(bytearray(msg), dtype=torch.uint8).to(dev), d_sig=torch.frombuffer(bytearray(sig), dtype=torch.uint8).to(dev),
            patches=sorted(patches.items())))
    d_st = torch.zeros(slots, dtype=torch.uint8, device=dev)
    d_root = torch.zeros(32 * slots, dtype=torch.uint8, device=dev)
    s_bls, s_mk = torch.cuda.Stream(), torch.cuda.Stream()

    def run(use_registry, overlapped):
        d_st.fill_(0xFF)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for slot, w in enumerate(work):
            st.patch(w["patches"])
            sb = s_bls.cuda_stream
            sm = s_mk.cuda_stream if overlapped else sb
            if use_registry:
                rc1 = L.ecgpu_fast_aggregate_verify_indexed_batch_dev(reg.handle, w["d_idx"].data_ptr(), w["d_off"].data_ptr(), w["k"],
                                                                      w["d_msg"].data_ptr(), w["d_sig"].data_ptr(), 1, 1,
                                                                      d_st.data_ptr() + slot, sb)
            else:
                rc1 = L.ecgpu_fast_aggregate_verify_batch_dev(w["d_keys"].data_ptr(), w["d_off"].data_ptr(), w["k"], w["d_msg"].data_ptr(),
                                                              w["d_sig"].data_ptr(), 1, 1, d_st.data_ptr() + slot, sb)
            rc2 = L.ecgpu_resident_state_root_dev(st.handle, d_root.data_ptr() + 32 * slot, sm)
            assert rc1 == 0 and rc2 == 0, L.ecgpu_last_error()
            if not overlapped:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert d_st.cpu().tolist() == [0] * slots
        return dt

    for use_registry in (False, True):
        for overlapped in (False, True):
            run(use_registry, overlapped)
            dt = min(run(use_registry, overlapped) for _ in range(2))
            print(f"sync aggregate ({'registry' if use_registry else 'reference semantics'}) + resident state root, "
                  f"{'two streams' if overlapped else 'one after the other'}: {1e3 * dt / slots:.2f} ms per slot = {slots / dt:.0f} slots/s",
                  flush=True)
    root_last = bytes(d_root[32 * (slots - 1):].cpu().numpy())
    for w in work * 3:  # the state after the runs above: every run applied the same patches
        for off, b in w["patches"]:
            enc[off:off + len(b)] = b
    print("last root equals the from-scratch root:", root_last == ssz.hash_tree_root_beacon_state_deneb(bytes(enc), 0), flush=True)
    st.close()


if __name__ == "__main__":
    main()
