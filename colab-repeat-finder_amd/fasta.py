"""Minimal FASTA reader standing in for the four things the reference's main() uses from
pyfastx.Fasta (reference perfect_repeat_finder.py:117,127,130,136-143): iteration in file order,
`name in fasta`, `fasta[name]`, and entries with `.name` / `.seq`.  Plain or gzip-compressed text.
Soft-masked (lower-case) sequence is returned as stored; the scan folds case itself."""
import gzip


class FastaEntry:
    __slots__ = ("name", "seq")

    def __init__(self, name, seq):
        self.name = name
        self.seq = seq

    def __len__(self):
        return len(self.seq)


class Fasta:
    def __init__(self, path):
        self.path = path
        self._entries = []
        self._by_name = {}
        opener = gzip.open if self._is_gzip(path) else open
        name, chunks = None, []
        with opener(path, "rt") as handle:
            for line in handle:
                if line.startswith(">"):
                    if name is not None:
                        self._add(name, chunks)
                    header = line[1:].strip()
                    name = header.split()[0] if header else ""
                    chunks = []
                elif name is not None:
                    chunks.append(line.strip())
        if name is not None:
            self._add(name, chunks)

    @staticmethod
    def _is_gzip(path):
        with open(path, "rb") as f:
            return f.read(2) == b"\x1f\x8b"

    def _add(self, name, chunks):
        entry = FastaEntry(name, "".join(chunks))
        self._entries.append(entry)
        self._by_name.setdefault(name, entry)

    def __iter__(self):
        return iter(self._entries)

    def __len__(self):
        return len(self._entries)

    def __contains__(self, name):
        return name in self._by_name

    def __getitem__(self, key):
        if isinstance(key, int):
            return self._entries[key]
        return self._by_name[key]
