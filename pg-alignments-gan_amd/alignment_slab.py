"""Packed input pipeline for the reference's on-disk training data.

The reference reads its alignments with two np.loadtxt calls per alignment and per batch
(DataGeneratorDisk.get_batch, src/train_wgan_v2.py:254-255; file format written by
src/sim_conversion/convert_ms.py:22-44 and convert_relative.py:37-90).  Here the same files are parsed once by
the library's native parser into a slab — 512 bytes of genotype bits and 64 fp32 positions per alignment —
that can be cached on disk, is moved to the GPU through pinned double buffers and expanded there into the
fp32 pool the training step reads (with the --normalize map fused).

Slab file: b"PGANSLAB", uint32 version (1), uint32 n, uint32 rows (64), uint32 cols (64), 8 reserved bytes,
then n x 512 bytes of bits (site j of individual i = bit j % 8 of byte i * 8 + j / 8), then n x 64 float32.
"""
from __future__ import annotations

import ctypes
import os
import struct
from typing import Sequence, Tuple

import numpy as np

MAGIC, VERSION, HEADER = b"PGANSLAB", 1, 32
ROWS = COLS = 64
BITS = ROWS * COLS // 8
_WHY = {1: "cannot be read", 2: "does not hold 64 x 64 sites / 64 positions", 3: "holds site values other than 0 and 1"}


class NotPackable(ValueError):
    """A sites file holds values other than 0 / 1 (e.g. label-smoothed data): use the float loader for it."""


def parse_pairs(sites_paths: Sequence[str], pos_paths: Sequence[str], threads: int = 8) -> Tuple[np.ndarray, np.ndarray]:
    """bits [n, 512] uint8 and positions [n, 64] float32 of n alignment pairs (native parser, `threads` threads).
    Positions are parsed as np.loadtxt(dtype=float32) parses them (decimal -> double -> float32)."""
    from pgan_hip import _lib
    n = len(sites_paths)
    if len(pos_paths) != n:
        raise ValueError("as many position files as sites files are needed")
    bits = np.zeros((n, BITS), np.uint8)
    pos = np.zeros((n, COLS), np.float32)
    if n == 0:
        return bits, pos
    status = np.zeros(n, np.int32)
    arr = ctypes.c_char_p * n
    sp = arr(*[os.fsencode(p) for p in sites_paths])
    pp = arr(*[os.fsencode(p) for p in pos_paths])
    failed = _lib.pgan_parse_alignments(ctypes.cast(sp, ctypes.c_void_p), ctypes.cast(pp, ctypes.c_void_p), n,
                                        bits.ctypes.data, pos.ctypes.data, status.ctypes.data, int(threads))
    if failed < 0:
        raise ValueError("pgan_parse_alignments rejected its arguments")
    if failed:
        k = int(np.flatnonzero(status)[0])
        msg = f"{sites_paths[k]} / {pos_paths[k]} {_WHY.get(int(status[k]), 'failed')} ({failed} of {n} pairs failed)"
        raise (NotPackable if int(status[k]) == 3 else ValueError)(msg)
    return bits, pos


def unpack_numpy(bits: np.ndarray) -> np.ndarray:
    """[n, 512] uint8 -> [n, 64, 64] float32 of 0 / 1 (host statement of the device kernel)."""
    return np.unpackbits(bits.reshape(-1, ROWS, COLS // 8), axis=2, bitorder="little").astype(np.float32)


def pack_numpy(sites: np.ndarray) -> np.ndarray:
    """[n, 64, 64] of 0 / 1 -> [n, 512] uint8."""
    return np.packbits(np.asarray(sites).reshape(-1, ROWS, COLS) >= 0.5, axis=2, bitorder="little").reshape(-1, BITS)


def write_slab(path: str, bits: np.ndarray, pos: np.ndarray) -> None:
    n = bits.shape[0]
    if bits.shape != (n, BITS) or bits.dtype != np.uint8 or pos.shape != (n, COLS) or pos.dtype != np.float32:
        raise ValueError("expected bits [n, 512] uint8 and positions [n, 64] float32")
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC + struct.pack("<IIII", VERSION, n, ROWS, COLS) + b"\0" * 8)
        f.write(np.ascontiguousarray(bits).tobytes())
        f.write(np.ascontiguousarray(pos).tobytes())
    os.replace(tmp, path)


def read_slab(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Memory-mapped (bits [n, 512], positions [n, 64]) of a slab file."""
    with open(path, "rb") as f:
        head = f.read(HEADER)
    if len(head) < HEADER or head[:8] != MAGIC:
        raise ValueError(f"{path} is not an alignment slab")
    version, n, rows, cols = struct.unpack("<IIII", head[8:24])
    if version != VERSION or rows != ROWS or cols != COLS:
        raise ValueError(f"{path}: unsupported slab (version {version}, {rows} x {cols})")
    if os.path.getsize(path) != HEADER + n * (BITS + 4 * COLS):
        raise ValueError(f"{path}: truncated slab")
    bits = np.memmap(path, np.uint8, "r", HEADER, (n, BITS))
    pos = np.memmap(path, np.float32, "r", HEADER + n * BITS, (n, COLS))
    return bits, pos


def to_device(bits: np.ndarray, pos: np.ndarray, device, normalize: bool = False, chunk: int = 8192):
    """(sites [n, 1, 64, 64], positions [n, 1, 64]) fp32 on `device`: the slab goes over in chunks through two
    pinned staging buffers (the copy of one chunk overlaps the expansion of the previous one) and is expanded by
    pgan_unpack_sites; normalize applies (x - 0.5) / 0.5 to both (src/train_wgan_v2.py:453,519-521)."""
    import torch
    from pgan_hip import _check, _lib, _stream
    n = bits.shape[0]
    sites = torch.empty(n, 1, ROWS, COLS, dtype=torch.float32, device=device)
    scale, shift = (2.0, -1.0) if normalize else (1.0, 0.0)
    if n:
        with torch.cuda.device(device):
            chunk = max(1, min(chunk, n))
            stage = [torch.empty(chunk, BITS, dtype=torch.uint8).pin_memory() for _ in range(2)]
            dbits = [torch.empty(chunk, BITS, dtype=torch.uint8, device=device) for _ in range(2)]
            done = [torch.cuda.Event(), torch.cuda.Event()]
            for i, lo in enumerate(range(0, n, chunk)):
                k, m = i & 1, min(chunk, n - lo)
                done[k].synchronize()                                  # staging buffer k is free again
                stage[k][:m].numpy()[...] = bits[lo:lo + m]
                dbits[k][:m].copy_(stage[k][:m], non_blocking=True)
                _check(_lib.pgan_unpack_sites(dbits[k].data_ptr(), sites[lo:].data_ptr(), m, scale, shift, _stream()),
                       "pgan_unpack_sites")
                done[k].record()
            torch.cuda.current_stream().synchronize()
    p = torch.from_numpy(np.array(pos, dtype=np.float32, copy=True)).to(device).view(n, 1, COLS)
    if normalize:
        p = p * 2.0 - 1.0
    return sites, p
