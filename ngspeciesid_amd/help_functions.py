"""FASTA/FASTQ reading and mkdir helper (reference modules/help_functions.py:13-53), written for this package."""
import errno
import os


def readfq(fp):
    """Yields (name, (seq, qual)) for FASTQ records and (name, (seq, None)) for FASTA records; multi-line records supported.
    Behaviour follows the reference's reader: '>'/'@' start a record, '+' starts the quality, qualities are read until
    they are as long as the sequence."""
    pending = None
    while True:
        if pending is None:
            for line in fp:
                if line[:1] in ">@":
                    pending = line[:-1] if line.endswith("\n") else line
                    break
            if pending is None:
                return
        name, chunks, pending_next = pending[1:], [], None
        for line in fp:
            if line[:1] in "@+>":
                pending_next = line[:-1] if line.endswith("\n") else line
                break
            chunks.append(line[:-1] if line.endswith("\n") else line)
        seq = "".join(chunks)
        if pending_next is None or pending_next[:1] != "+":
            yield name, (seq, None)
            if pending_next is None:
                return
            pending = pending_next
            continue
        qual_chunks, got, done = [], 0, False
        for line in fp:
            piece = line[:-1] if line.endswith("\n") else line
            qual_chunks.append(piece); got += len(piece)
            if got >= len(seq):
                done = True
                break
        if done:
            yield name, (seq, "".join(qual_chunks))
            pending = None
        else:
            yield name, (seq, None)
            return


def mkdir_p(path):
    try:
        os.makedirs(path)
    except OSError as exc:
        if not (exc.errno == errno.EEXIST and os.path.isdir(path)):
            raise
