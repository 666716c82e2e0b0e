"""FlatFile: format parity with the reference's writer, the reference's access surface,
and the zero-copy packed-batch path into the encoder."""
import os

import numpy as np
import pytest

EXPECT = [b"MKVLAAGIVGLLLAQPSNA", b"", b"ACDEFGHIKLMNPQRSTVWYacdef", b"ACGTN", b"WWWW"]


def test_reads_the_reference_written_file(golden_dir):
    from bioseq_amd.flatfile import FlatFile
    ff = FlatFile(os.path.join(golden_dir, "flatfile_small.ff"))
    assert len(ff) == ff.nseqs() == ff.size() == 5 and ff.maxseqlen == ff.max_seq_len == 25
    assert ff.seq_offset() == (5 + 2) * 8
    assert ff.indptr().dtype == np.uint64 and ff.indptr().tolist() == [0, 19, 19, 44, 49, 53]
    assert [bytes(x) for x in ff] == EXPECT
    assert isinstance(ff[0], bytearray) and bytes(ff[-1]) == b"WWWW" and bytes(ff.access(3)) == b"ACGTN"
    assert [bytes(x) for x in ff.access(0, 3)] == EXPECT[:3] and [bytes(x) for x in ff[1:4]] == EXPECT[1:4]
    assert [bytes(x) for x in ff.access(0, 5, 2)] == EXPECT[::2]
    assert [bytes(x) for x in ff[np.array([4, 0])]] == [EXPECT[4], EXPECT[0]]
    with pytest.raises(IndexError):
        ff.access(5)
    chars, offs = ff.packed(2, 5)
    assert offs.tolist() == [0, 25, 30, 34] and chars.tobytes() == b"".join(EXPECT[2:5])


def test_writer_is_byte_identical_to_the_reference(golden_dir, tmp_path):
    from bioseq_amd.flatfile import FlatFile
    out = str(tmp_path / "mine.ff")
    ff = FlatFile(os.path.join(golden_dir, "flatfile_small.fa"), out)
    assert ff.path == out
    assert open(out, "rb").read() == open(os.path.join(golden_dir, "flatfile_small.ff"), "rb").read()
    # gzip input and the default output name
    import gzip
    import shutil
    gz = str(tmp_path / "x.fa.gz")
    with open(os.path.join(golden_dir, "flatfile_small.fa"), "rb") as a, gzip.open(gz, "wb") as b:
        shutil.copyfileobj(a, b)
    ff2 = FlatFile(gz, "")
    assert ff2.path == gz + ".ff" and [bytes(x) for x in ff2] == EXPECT


@pytest.mark.gpu
def test_flatfile_to_encoder_zero_copy(gpu, bsq, oracle, golden_dir, tmp_path):
    from bioseq_amd import synth
    from bioseq_amd.flatfile import FlatFile, write_flatfile
    chars, offs = synth.synth_packed(77, 3000, 0, 300, synth.AA)
    path = write_flatfile(synth.unpack(chars, offs), str(tmp_path / "big.ff"))
    ff = FlatFile(path)
    tok, ora = bsq.Tokenizer("AMINO20", 1, 1, 0), oracle.OracleTokenizer("AMINO20", 1, 1, 0)
    P = ff.maxseqlen + 2
    for a, b in ((0, 3000), (100, 1124), (2990, 3000)):
        c, o = synth.unpack(chars, offs)[a:b], None
        exp_t = ora.batch_tokenize(c, padlen=P, batch_first=True)
        exp_o = ora.batch_onehot_encode(c, padlen=P, destchar="f")
        assert ff.batch_tokenize(tok, a, b).tobytes() == exp_t.tobytes()                         # mapped file -> staged
        assert ff.batch_tokenize(tok, a, b, device=gpu).cpu().numpy().tobytes() == exp_t.tobytes()  # resident store
        assert ff.batch_onehot_encode(tok, a, b, destchar="f", device=gpu).cpu().numpy().tobytes() == exp_o.tobytes()
        # same result as the reference's route: access() -> list of bytearrays -> batch_tokenize
        assert tok.batch_tokenize(ff.access(a, b), padlen=P, batch_first=True).tobytes() == exp_t.tobytes()
