"""Synthetic input generators (numpy side)."""
import numpy as np


def test_single_shape_and_content(pkg):
    from fastqandfurious_amd import synth
    a = synth.single(0, 5, seed=42)
    assert a.size == 5 * 322
    rec = a.reshape(5, 322)
    assert bytes(rec[3, :18]) == b"@SYN.0000000003/1\n"
    assert (rec[:, 168] == 10).all() and (rec[:, 169] == 43).all() and (rec[:, 321] == 10).all()
    assert set(np.unique(rec[:, 18:168])) <= set(b"ACGT")
    q = rec[:, 171:321]
    assert q.min() >= 33 and q.max() <= 73
    # counter based: a slice equals the same records generated alone
    b = synth.single(3, 2, seed=42)
    assert (b == a[3 * 322:]).all()
    assert synth.single_records_for(1 << 30) == 3334601


def test_wrapped_sizes(pkg):
    from fastqandfurious_amd import synth
    data, start = synth.wrapped(10, 40, seed=43)
    assert start[-1] == data.size
    sizes = synth.wrapped_sizes(10, 40, seed=43)
    assert (np.diff(start) == sizes).all()
    b = bytes(data)
    for i in range(40):
        rec = b[start[i]:start[i + 1]]
        assert rec.startswith(b"@SYN.%010d/1\n" % (10 + i)) and rec.endswith(b"\n")
        assert max(len(x) for x in rec.split(b"\n")) <= 80
    d2, s2 = synth.wrapped(20, 5, seed=43)
    assert bytes(d2) == b[start[10]:start[15]]
