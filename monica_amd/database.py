"""Host-side mirror of `monica.genomes.database` (reference: monica/genomes/database.py): the
step that turns downloaded genome FASTA files into the `databaseN.fna.gz` chunks which
`aligner.indexer` (aligner.py:31-53) then indexes.

Same names, argument order, return values, printed messages and files.  The one thing the
aligner depends on is the header contract of `builder` (database.py:59-62): every record of a
genome is renamed to `<tax_unit>:<accession>`, so all contigs of a genome share one contig name
and `best[0].split(':')` (aligner.py:234, 240) recovers taxon and accession.  Biopython is not
part of this image; FASTA records are read and written here the way `SeqIO.parse` /
`SeqIO.write` do it (title = new id + ' ' + old title, sequence wrapped at 60 columns).
"""
import gzip
import itertools
import os
import pickle
from multiprocessing.dummy import Pool as ThreadPool

from .aligner import GENOMES_PATH

DATABASES_PATH = os.path.join(GENOMES_PATH, "databases") if GENOMES_PATH else None
DATABASE_NAME = ["database", ".fna.gz"]
_LENGTHS_FILE = "current_genomes_length.pkl"
_WRAP = 60                                          # Biopython's FastaWriter line width


def _gz_files(folder):
    return [os.path.join(folder, f) for f in os.listdir(folder) if f.endswith(".fna.gz")]


def multi_threaded_builder(genomes=None, max_chunk_size=None, databases_path=DATABASES_PATH,
                           database_name=DATABASE_NAME, keep_genomes=None, n_threads=None):
    """database.py:16-49.  `genomes` = [(path to <genome>.fna.gz, (tax_unit, accession)), ...];
    one database file per size-bounded chunk, written by a pool of threads.  Returns
    (databases_path, {accession: genome length}) and persists the lengths for `normalizer`."""
    if os.path.exists(databases_path):
        for stale in _gz_files(databases_path):     # a rebuild starts from an empty folder
            os.remove(stale)
    else:
        os.makedirs(databases_path)

    lengths_path = os.path.join(GENOMES_PATH, _LENGTHS_FILE)
    known = {}
    if os.path.exists(lengths_path):
        with open(lengths_path, "rb") as handle:
            known = pickle.load(handle)

    jobs = zip(_genomes_splitter(genomes, max_chunk_size=max_chunk_size), itertools.repeat(databases_path),
               itertools.repeat(database_name), itertools.count())
    pool = ThreadPool(n_threads)
    try:
        for chunk_lengths in pool.starmap(builder, jobs):
            known.update(chunk_lengths)
    finally:
        pool.close()

    if not keep_genomes:                            # the downloads are not kept once they are in a database
        for downloaded in _gz_files(GENOMES_PATH):
            os.remove(downloaded)
    with open(lengths_path, "wb") as handle:
        pickle.dump(known, handle)
    open(os.path.join(GENOMES_PATH, "database_created"), "wb").close()      # progress marker
    return databases_path, known


def _fasta_records(handle):
    """(title, sequence) pairs as SeqIO.parse(handle, 'fasta') would give them."""
    title, parts = None, []
    for line in handle:
        if line.startswith(">"):
            if title is not None:
                yield title, "".join(parts)
            title, parts = line[1:].rstrip(), []
        elif title is not None:
            parts.append("".join(line.split()))
    if title is not None:
        yield title, "".join(parts)


def _retitled(new_id, old_title):
    """Title SeqIO.write gives a record after `seq_record.id = new_id`: the old title stays on
    as the description."""
    if not old_title:
        return new_id
    return old_title if old_title.split(None, 1)[0] == new_id else new_id + " " + old_title


def builder(genomes_chunk, databases_path, database_name, database_number):
    """database.py:52-67: concatenate one chunk under `tax_unit:accession` headers; returns
    {accession: bases in that genome}."""
    file_name = str(database_number).join(database_name)
    print("Working on {}".format(file_name))
    lengths = {}
    with gzip.open(os.path.join(databases_path, file_name), "wt") as out:
        for path, (tax_unit, accession) in ((g[0], g[1]) for g in genomes_chunk):
            header = ":".join((tax_unit, accession))
            total = 0
            with gzip.open(path, "rt") as genome:
                for title, seq in _fasta_records(genome):
                    total += len(seq)
                    out.write(">{}\n".format(_retitled(header, title)))
                    out.writelines(seq[i:i + _WRAP] + "\n" for i in range(0, len(seq), _WRAP))
            lengths[accession] = total
    print("Finished building {}".format(file_name))
    return lengths


def _genomes_splitter(genomes, max_chunk_size=None):
    """database.py:70-92: chunks whose compressed sizes add up to at most `max_chunk_size`; a
    genome larger than the limit is announced and goes alone.  As in the reference, the genome
    that closes a chunk by not fitting into it is NOT carried into the next chunk."""
    pending, pending_size = [], 0
    for genome in genomes:
        size = os.path.getsize(genome[0])
        if size > max_chunk_size:
            print("Genome {}, ({}) alone expected to generate an index "
                  "exceeding the maximum memory deriving from settings of {} bytes"
                  .format(genome[0], genome[1][0], (size - max_chunk_size) * 16))
            yield [genome]
        elif pending_size + size <= max_chunk_size:
            pending.append(genome)
            pending_size += size
        else:
            yield pending
            pending, pending_size = [], 0
    if pending:
        yield pending
