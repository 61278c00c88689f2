"""Minimal FASTQ reader / writer for the aligner's host side.

The reference iterates `SeqIO.parse(sample, 'fastq')` and writes with `SeqIO.write(record,
handle, 'fastq')` (monica/genomes/aligner.py:191, 212, 232-243, 265).  Biopython is not part
of this image, and per-record Python objects are what keeps the reference's loop slow, so the
reader here returns whole batches as flat arrays and the writer reproduces Biopython's title
rule: the header is the original description line unless the record id was replaced, in
which case it is `<new id> <original description>`.
"""
import numpy as np


class FastqBatch:
    """n reads as flat arrays: `bases` (uint8, concatenated), `offsets` (int64[n+1]),
    plus per-read header / quality bytes for re-emission."""
    __slots__ = ("ids", "headers", "quals", "bases", "offsets")

    def __init__(self, ids, headers, seqs, quals):
        self.ids = ids
        self.headers = headers
        self.quals = quals
        self.offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
        if seqs:
            np.cumsum([len(s) for s in seqs], out=self.offsets[1:])
        self.bases = np.frombuffer(b"".join(seqs), dtype=np.uint8) if seqs else np.zeros(0, dtype=np.uint8)

    def __len__(self):
        return len(self.ids)

    def seq(self, r):
        return self.bases[self.offsets[r]:self.offsets[r + 1]].tobytes()


def read_batches(path, max_reads=100_000, max_bases=1 << 29):
    """Yield FastqBatch objects from a 4-line-per-record FASTQ file (sequence and quality may
    also be wrapped over several lines, as Biopython accepts)."""
    ids, headers, seqs, quals = [], [], [], []
    n_bases = 0
    with open(path, "rb") as f:
        line = f.readline()
        while line:
            if not line.strip():
                line = f.readline()
                continue
            if not line.startswith(b"@"):
                raise ValueError("Records in Fastq files should start with '@' character")
            header = line[1:].rstrip(b"\r\n")
            seq_parts = []
            line = f.readline()
            while line and not line.startswith(b"+"):
                seq_parts.append(line.strip())
                line = f.readline()
            if not line:
                raise ValueError("End of file without quality information.")
            seq = b"".join(seq_parts)
            qual_parts, q_len = [], 0
            line = f.readline()
            while line and q_len < len(seq):
                qual_parts.append(line.strip())
                q_len += len(qual_parts[-1])
                line = f.readline()
            qual = b"".join(qual_parts)
            if len(qual) != len(seq):
                raise ValueError("Lengths of sequence and quality values differs for %s (%i and %i)."
                                 % (header.decode(errors="replace"), len(seq), len(qual)))
            hd = header.decode(errors="replace")
            ids.append(hd.split(None, 1)[0] if hd.strip() else "")
            headers.append(hd)
            seqs.append(seq)
            quals.append(qual)
            n_bases += len(seq)
            if len(ids) >= max_reads or n_bases >= max_bases:
                yield FastqBatch(ids, headers, seqs, quals)
                ids, headers, seqs, quals, n_bases = [], [], [], [], 0
    if ids:
        yield FastqBatch(ids, headers, seqs, quals)


def format_record(batch, r, new_id=None):
    """One record as bytes.  `new_id` mimics `seq_record.id = tax_unit` (aligner.py:242)."""
    header = batch.headers[r]
    if new_id is not None and (not header or header.split(None, 1)[0] != new_id):
        title = f"{new_id} {header}" if header else new_id
    else:
        title = header
    return b"@" + title.encode() + b"\n" + batch.seq(r) + b"\n+\n" + batch.quals[r] + b"\n"
