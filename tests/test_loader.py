"""Native alignment parser and slab format against the reference's loader arithmetic (np.loadtxt of
`<n>_sites.csv` / `<n>_pos.csv`, src/train_wgan_v2.py:254-255): bit-exact values, round trips, and the
error paths (missing file, ragged rows, non-binary sites)."""
import os

import numpy as np
import pytest

import dcwgan_oracle as orc


def _write(tmp, n, seed=5):
    rs, rp = orc.synthetic_alignments(n, seed)
    sp, pp = [], []
    for i in range(n):
        sp.append(os.path.join(tmp, f"{i}_sites.csv"))
        pp.append(os.path.join(tmp, f"{i}_pos.csv"))
        np.savetxt(sp[-1], rs[i, 0].numpy().astype(int), fmt="%d", delimiter=",")
        np.savetxt(pp[-1], rp[i, 0].numpy(), fmt="%.6f")
    return sp, pp


def test_parser_equals_loadtxt(tmp_path):
    import alignment_slab as sl
    sp, pp = _write(str(tmp_path), 40)
    bits, pos = sl.parse_pairs(sp, pp, threads=3)
    want_s = np.stack([np.loadtxt(a, delimiter=",", dtype=np.float32) for a in sp])
    want_p = np.stack([np.loadtxt(a, dtype=np.float32) for a in pp])
    assert np.array_equal(sl.unpack_numpy(bits), want_s)
    assert np.array_equal(pos, want_p)
    assert np.array_equal(sl.pack_numpy(want_s), bits)
    one_b, one_p = sl.parse_pairs(sp[:1], pp[:1], threads=1)
    assert np.array_equal(one_b, bits[:1]) and np.array_equal(one_p, pos[:1])
    empty_b, empty_p = sl.parse_pairs([], [])
    assert empty_b.shape == (0, 512) and empty_p.shape == (0, 64)


def test_parser_accepts_loadtxt_spellings(tmp_path):
    """Float spellings, blank lines, CRLF and a missing final newline parse as np.loadtxt parses them; positions
    go decimal -> double -> float32."""
    import alignment_slab as sl
    rng = np.random.default_rng(3)
    s = (rng.random((64, 64)) < 0.3).astype(int)
    p = np.sort(rng.random(64))
    sf, pf = str(tmp_path / "a_sites.csv"), str(tmp_path / "a_pos.csv")
    rows = [",".join(("1.0" if v else "0") if j % 3 else ("1e0" if v else "0.000") for j, v in enumerate(r)) for r in s]
    with open(sf, "w", newline="") as f:
        f.write("\r\n".join(rows[:10]) + "\r\n\r\n" + "\n".join(rows[10:]))
    with open(pf, "w") as f:
        f.write("\n".join(repr(float(v)) for v in p))
    bits, pos = sl.parse_pairs([sf], [pf])
    assert np.array_equal(sl.unpack_numpy(bits)[0], np.loadtxt(sf, delimiter=",", dtype=np.float32))
    assert np.array_equal(pos[0], np.loadtxt(pf, dtype=np.float32))


def test_parser_errors(tmp_path):
    import alignment_slab as sl
    sp, pp = _write(str(tmp_path), 3)
    with pytest.raises(ValueError, match="cannot be read"):
        sl.parse_pairs(sp + [str(tmp_path / "nope_sites.csv")], pp + [pp[0]])
    ragged = str(tmp_path / "r_sites.csv")
    lines = open(sp[0]).read().splitlines()
    open(ragged, "w").write("\n".join(lines[:30] + [lines[30][:-2]] + lines[31:]) + "\n")
    with pytest.raises(ValueError, match="64 x 64"):
        sl.parse_pairs([ragged], [pp[0]])
    short = str(tmp_path / "s_pos.csv")
    open(short, "w").write("0.1\n0.2\n")
    with pytest.raises(ValueError, match="64 positions"):
        sl.parse_pairs([sp[0]], [short])
    smooth = str(tmp_path / "m_sites.csv")
    open(smooth, "w").write(open(sp[0]).read().replace("1", "0.95", 1))
    with pytest.raises(sl.NotPackable):
        sl.parse_pairs([smooth], [pp[0]])
    with pytest.raises(ValueError):
        sl.parse_pairs(sp, pp[:2])


def test_slab_round_trip_and_validation(tmp_path):
    import alignment_slab as sl
    sp, pp = _write(str(tmp_path), 12)
    bits, pos = sl.parse_pairs(sp, pp)
    path = str(tmp_path / "x.slab")
    sl.write_slab(path, bits, pos)
    assert os.path.getsize(path) == 32 + 12 * 768
    b2, p2 = sl.read_slab(path)
    assert np.array_equal(b2, bits) and np.array_equal(p2, pos)
    with open(path, "r+b") as f:
        f.truncate(32 + 12 * 768 - 4)
    with pytest.raises(ValueError, match="truncated"):
        sl.read_slab(path)
    open(path, "wb").write(b"not a slab at all, but long enough to hold a header")
    with pytest.raises(ValueError, match="not an alignment slab"):
        sl.read_slab(path)
    with pytest.raises(ValueError):
        sl.write_slab(path, bits.astype(np.int32), pos)


def test_disk_loader_uses_the_parser_and_falls_back(tmp_path):
    """DataGeneratorDisk.load_all: packed route for 0 / 1 files, np.loadtxt route for label-smoothed ones; both
    return what the reference's per-file np.loadtxt returns."""
    import random

    import train_wgan_v2 as tw
    sp, pp = _write(str(tmp_path), 10)
    random.seed(4)
    d = tw.DataGeneratorDisk(str(tmp_path), num_in=10, batch_size=2)
    s, p = d.load_all()
    for i, u in enumerate(d.indices):
        assert np.array_equal(s[i, 0].numpy(), np.loadtxt(d.sites[u], delimiter=",", dtype=np.float32))
        assert np.array_equal(p[i, 0].numpy(), np.loadtxt(d.positions[u], dtype=np.float32))
    text = open(sp[3]).read().replace("1", "0.95", 1)
    with open(sp[3], "w") as f:
        f.write(text)
    s2, _ = d.load_all()
    k = d.indices.index(d.sites.index(sp[3]))
    assert float(s2[k].max()) == 1.0 and bool((s2[k] == np.float32(0.95)).any())


@pytest.mark.gpu
@pytest.mark.parametrize("n,normalize", [(1, False), (37, True), (300, False)])
def test_device_unpack_and_pack(n, normalize):
    import torch

    import alignment_slab as sl
    from pgan_hip import _check, _lib, _stream
    rng = np.random.default_rng(n)
    bits = rng.integers(0, 256, (n, 512), dtype=np.uint8)
    pos = np.sort(rng.random((n, 64)).astype(np.float32), axis=1)
    dev = torch.device("cuda", 0)
    s, p = sl.to_device(bits, pos, dev, normalize=normalize, chunk=128)
    want = sl.unpack_numpy(bits)
    if normalize:
        want, pos = want * 2 - 1, pos * 2 - 1
    assert np.array_equal(s.cpu().numpy()[:, 0], want)
    assert np.array_equal(p.cpu().numpy()[:, 0], pos)
    back = torch.zeros(n, 512, dtype=torch.uint8, device=dev)
    x = torch.from_numpy(sl.unpack_numpy(bits)).to(dev).contiguous()
    _check(_lib.pgan_pack_sites(x.data_ptr(), back.data_ptr(), n, _stream()), "pgan_pack_sites")
    assert np.array_equal(back.cpu().numpy(), bits)
