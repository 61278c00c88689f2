"""FASTQ batches for the aligner's host side.

The reference iterates `SeqIO.parse(sample, 'fastq')` and writes with `SeqIO.write(record,
handle, 'fastq')` (monica/genomes/aligner.py:191, 212, 232-243, 265).  Biopython is not part
of this image, and per-record Python objects are what keeps the reference's loop slow: the
parser and the writer live behind the C-ABI (`mnc_fastq_*`, monica_amd/csrc/hostio.cpp) and
work on whole batches.  This module is the thin Python face of it: `read_batches` yields
snapshots of the library's batches for code that wants to look at single records, and
`format_record` states Biopython's title rule -- the header is the original title unless the
record id was replaced, in which case it is `<new id> <original title>`.
"""
import numpy as np

from . import _capi


class FastqBatch:
    """n reads as flat arrays: `bases` / `quals` (uint8, concatenated), `offsets` (int64[n+1]),
    plus the title lines (`headers`) and their first words (`ids`)."""
    __slots__ = ("ids", "headers", "quals", "bases", "offsets")

    def __init__(self, headers, bases, quals, offsets):
        self.headers = headers
        self.ids = [h.split(None, 1)[0] if h.strip() else "" for h in headers]
        self.bases, self.quals, self.offsets = bases, quals, offsets

    @classmethod
    def from_reader(cls, reader):
        return cls([reader.title(r) for r in range(reader.n)], reader.bases().copy(), reader.quals().copy(),
                   reader.offsets().copy())

    def __len__(self):
        return len(self.headers)

    def seq(self, r):
        return self.bases[self.offsets[r]:self.offsets[r + 1]].tobytes()

    def qual(self, r):
        return self.quals[self.offsets[r]:self.offsets[r + 1]].tobytes()


def read_batches(path, max_reads=100_000, max_bases=1 << 29):
    """Yield FastqBatch snapshots of a FASTQ file (sequence and quality may be wrapped over
    several lines, as Biopython accepts); ValueError with Biopython's message when malformed."""
    reader = _capi.FastqReader(path)
    try:
        while reader.next(max_reads, max_bases):
            yield FastqBatch.from_reader(reader)
    finally:
        reader.close()


def format_record(batch, r, new_id=None):
    """One record as bytes.  `new_id` mimics `seq_record.id = tax_unit` (aligner.py:242)."""
    header = batch.headers[r]
    if new_id is not None and (not header or header.split(None, 1)[0] != new_id):
        title = f"{new_id} {header}" if header else new_id
    else:
        title = header
    return b"@" + title.encode() + b"\n" + batch.seq(r) + b"\n+\n" + batch.qual(r) + b"\n"
