"""Host-side mirror of `monica.genomes.database` (reference: monica/genomes/database.py): the
step that turns downloaded genome FASTA files into the `databaseN.fna.gz` chunks which
`aligner.indexer` (aligner.py:31-53) then indexes.

Same names, argument order, return values and files.  The one thing the aligner depends on is
the header contract of `builder` (database.py:59-62): every record of a genome is renamed to
`<tax_unit>:<accession>`, so all contigs of a genome share one contig name and
`best[0].split(':')` (aligner.py:234, 240) recovers taxon and accession.  Biopython is not part
of this image; FASTA records are read and written here the way `SeqIO.parse` / `SeqIO.write`
do it (title = new id + ' ' + old title, sequence wrapped at 60 columns).
"""
import gzip
import os
import pickle
from itertools import count, repeat
from multiprocessing.dummy import Pool as ThreadPool

from .aligner import GENOMES_PATH

DATABASES_PATH = os.path.join(GENOMES_PATH, "databases") if GENOMES_PATH else None
DATABASE_NAME = ["database", ".fna.gz"]


def multi_threaded_builder(genomes=None, max_chunk_size=None, databases_path=DATABASES_PATH,
                           database_name=DATABASE_NAME, keep_genomes=None, n_threads=None):
    """Chunk the genomes by compressed size and write one database file per chunk
    (database.py:16-49).  Returns (databases_path, {accession: genome length})."""
    if not os.path.exists(databases_path):
        os.makedirs(databases_path)
    else:
        for database in os.listdir(databases_path):
            if database.endswith(".fna.gz"):
                os.remove(os.path.join(databases_path, database))

    lengths_file = os.path.join(GENOMES_PATH, "current_genomes_length.pkl")
    if os.path.exists(lengths_file):
        with open(lengths_file, "rb") as f:
            current_genomes_length = pickle.load(f)
    else:
        current_genomes_length = dict()

    pool = ThreadPool(n_threads)
    try:
        lengths = pool.starmap(builder, zip(_genomes_splitter(genomes, max_chunk_size=max_chunk_size),
                                            repeat(databases_path), repeat(database_name), count()))
    finally:
        pool.close()
    for length in lengths:
        current_genomes_length.update(length)

    if not keep_genomes:
        for genome in os.listdir(GENOMES_PATH):
            if genome.endswith(".fna.gz"):
                os.remove(os.path.join(GENOMES_PATH, genome))

    with open(lengths_file, "wb") as f:
        pickle.dump(current_genomes_length, f)
    with open(os.path.join(GENOMES_PATH, "database_created"), "wb"):
        pass
    return databases_path, current_genomes_length


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


def builder(genomes_chunk, databases_path, database_name, database_number):
    """Concatenate one chunk of genomes under the `tax_unit:accession` headers (database.py:52-67)."""
    database_file = os.path.join(databases_path, str(database_number).join(database_name))
    print("Working on {}".format(str(database_number).join(database_name)))
    this_database_genomes_length = dict()
    with gzip.open(database_file, "wt") as database:
        for genome in genomes_chunk:
            genome_length = 0
            new_header = ":".join(genome[1])
            with gzip.open(genome[0], "rt") as g:
                for title, seq in _fasta_records(g):
                    genome_length += len(seq)
                    # SeqIO.write after `seq_record.id = new_header`: the old title stays as description
                    if title and title.split(None, 1)[0] == new_header:
                        header = title
                    elif title:
                        header = "{} {}".format(new_header, title)
                    else:
                        header = new_header
                    database.write(">" + header + "\n")
                    for i in range(0, len(seq), 60):
                        database.write(seq[i:i + 60] + "\n")
            this_database_genomes_length[genome[1][1]] = genome_length
    print("Finished building {}".format(str(database_number).join(database_name)))
    return this_database_genomes_length


def _genomes_splitter(genomes, max_chunk_size=None):
    """Chunks of genomes whose compressed sizes add up to at most `max_chunk_size`; a genome
    larger than that goes alone (database.py:70-92).  As in the reference, the genome that
    closes a chunk by not fitting into it is not carried into the next chunk."""
    chunk = []
    exceeding_chunk = []
    chunk_size = 0
    for genome in genomes:
        size = os.path.getsize(genome[0])
        if size > max_chunk_size:
            exceeding_chunk.append(genome)
            print("Genome {}, ({}) alone expected to generate an index "
                  "exceeding the maximum memory deriving from settings of {} bytes"
                  .format(genome[0], genome[1][0], (size - max_chunk_size) * 16))
            yield exceeding_chunk
            exceeding_chunk = []
        else:
            if chunk_size + size <= max_chunk_size:
                chunk.append(genome)
                chunk_size += size
            else:
                yield chunk
                chunk = []
                chunk_size = 0
    if chunk:
        yield chunk
