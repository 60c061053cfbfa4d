"""Random values of oracle/ssz.py types (test helper): list lengths are drawn small with occasional full / empty lists."""
from oracle import ssz


def random_value(t, r, fill=None):
    if isinstance(t, ssz.UInt):
        return r.randrange(1 << t.bits)
    if isinstance(t, ssz.Boolean):
        return r.random() < 0.5
    if isinstance(t, ssz.ByteVector):
        return r.randbytes(t.n)
    if isinstance(t, ssz.ByteList):
        return r.randbytes(_length(t.limit, r, fill, cap=300))
    if isinstance(t, ssz.Bitvector):
        return [r.random() < 0.5 for _ in range(t.n)]
    if isinstance(t, ssz.Bitlist):
        return [r.random() < 0.5 for _ in range(_length(t.limit, r, fill, cap=2048))]
    if isinstance(t, ssz.Vector):
        return [random_value(t.elem, r, fill) for _ in range(t.n)]
    if isinstance(t, ssz.SSZList):
        return [random_value(t.elem, r, fill) for _ in range(_length(t.limit, r, fill, cap=40))]
    if isinstance(t, ssz.Container):
        return {n: random_value(ft, r, fill) for n, ft in t.fields}
    raise TypeError(t)


def _length(limit, r, fill, cap):
    if fill == "empty":
        return 0
    if fill == "full":
        return min(limit, cap)
    return min(limit, r.choice([0, 1, 2, 3, r.randrange(1, cap + 1)]))
