"""Byte-level host model of a BeaconState encoding (fixed part + variable-size fields) kept in step with a resident state:
splices, offset words and patch positions are computed here from the ORACLE's container type, independently of csrc/state_plan.h."""


class EncodingModel:
    def __init__(self, container, enc: bytes):
        self.words = {}
        pos = 0
        self.fixed_ranges = {}
        for name, ty in container.fields:
            if ty.fixed_size is None:
                self.words[name] = pos
                pos += 4
            else:
                self.fixed_ranges[name] = (pos, pos + ty.fixed_size)
                pos += ty.fixed_size
        self.fixed_len = pos
        self.names = list(self.words)
        offs = [int.from_bytes(enc[self.words[n]:self.words[n] + 4], "little") for n in self.names] + [len(enc)]
        assert offs[0] == pos
        self.fixed = bytearray(enc[:pos])
        self.var = {n: bytearray(enc[offs[i]:offs[i + 1]]) for i, n in enumerate(self.names)}

    def start(self, name):
        off = self.fixed_len
        for n in self.names:
            if n == name:
                return off
            off += len(self.var[n])
        raise KeyError(name)

    def encoding(self) -> bytes:
        off = self.fixed_len
        for n in self.names:
            self.fixed[self.words[n]:self.words[n] + 4] = off.to_bytes(4, "little")
            off += len(self.var[n])
        return bytes(self.fixed) + b"".join(bytes(self.var[n]) for n in self.names)

    def write(self, off: int, data: bytes):
        """the same overwrite a patch performs, addressed in the CURRENT encoding"""
        if off + len(data) <= self.fixed_len:
            self.fixed[off:off + len(data)] = data
            return
        assert off >= self.fixed_len
        for n in self.names:
            s = self.start(n)
            if s <= off and off + len(data) <= s + len(self.var[n]):
                self.var[n][off - s:off - s + len(data)] = data
                return
        raise ValueError("a patch must stay inside one field here")
