"""The reference's video file hand-off without imageio / torchvision / ffmpeg: Motion-JPEG in an ISO base media
file (.mp4).

The reference's scripts end in `imageio.mimwrite(path, videos[0], fps=8, quality=9)` (base/pipelines/sample.py:91, vsr/sample.py:140)
or `torchvision.io.write_video(path, video_, fps=fps)` (interpolation/sample.py:299), and the VSR script starts from
`torchvision.io.read_video(filename, pts_unit='sec', output_format='TCHW')` (vsr/sample.py:85).  Those calls sit on an ffmpeg
build (H.264), which this image does not have and which is far outside the hot path.  What a caller of this package needs is a
playable file per prompt and a way to read it back for the next stage, so this module writes every frame as a baseline JPEG
(Pillow) into one `mdat` box and describes them with the sample tables of a single video track: sample entry `mp4v` with an
`esds` whose objectTypeIndication is 0x6C (ISO/IEC 10918-1 JPEG) — the layout ffmpeg itself writes for `-c:v mjpeg out.mp4`, which
ffplay, VLC and browsers' media stacks read.  `read_video` parses exactly this layout back (it is not an H.264 decoder: a file
from the reference's own writers is refused with a message saying so).

Same call shapes as the reference's two writers and its reader:
    mimwrite(uri, ims, fps=8, quality=9)                      # imageio.mimwrite: ims [T, H, W, 3] uint8, quality 0..10
    write_video(filename, video_array, fps)                    # torchvision.io.write_video: [T, H, W, 3] uint8 tensor
    read_video(filename, pts_unit='sec', output_format='THWC') # torchvision.io.read_video -> (vframes, aframes, info)
"""
import io
import struct
from fractions import Fraction
from typing import List, Tuple

import numpy as np
import torch

_MATRIX = struct.pack(">9i", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def _box(kind: bytes, *payload: bytes) -> bytes:
    body = b"".join(payload)
    return struct.pack(">I4s", 8 + len(body), kind) + body


def _full(kind: bytes, version: int, flags: int, *payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", (version << 24) | flags), *payload)


def _descr(tag: int, body: bytes) -> bytes:
    assert len(body) < 128          # one-byte size field is enough for every descriptor written here
    return bytes([tag, len(body)]) + body


def _frames_uint8(ims) -> np.ndarray:
    if isinstance(ims, torch.Tensor):
        ims = ims.detach().cpu().numpy()
    arr = np.asarray(ims)
    if arr.ndim != 4 or arr.shape[-1] != 3:
        raise ValueError(f"expected frames [T, H, W, 3], got {arr.shape}")
    if arr.dtype != np.uint8:
        raise ValueError(f"expected uint8 frames (the pipelines' `.video` output), got {arr.dtype}")
    if arr.shape[0] == 0:
        raise ValueError("no frames")
    return np.ascontiguousarray(arr)


def _jpeg_quality(quality: float) -> int:
    """imageio's 0..10 scale (10 best; the reference passes 9) -> Pillow's 1..95."""
    if not 0 <= quality <= 10:
        raise ValueError("quality must be in 0..10")
    return int(min(95, max(5, round(5 + 9 * quality))))


def mimwrite(uri, ims, fps: float = 8, quality: float = 9) -> None:
    """imageio.mimwrite(uri, ims, fps=..., quality=...) for an .mp4 target (base/pipelines/sample.py:91, vsr/sample.py:140)."""
    from PIL import Image
    frames = _frames_uint8(ims)
    n, h, w, _ = frames.shape
    q = _jpeg_quality(quality)
    rate = Fraction(fps).limit_denominator(1001)
    timescale = rate.numerator * 1000 if rate.numerator < 1000 else rate.numerator         # ticks per second
    delta = timescale * rate.denominator // rate.numerator                                   # ticks per frame
    samples = []
    for f in frames:
        buf = io.BytesIO()
        Image.fromarray(f, "RGB").save(buf, format="JPEG", quality=q, subsampling=0 if q >= 90 else 2, optimize=False)
        samples.append(buf.getvalue())
    duration = delta * n

    ftyp = _box(b"ftyp", b"isom", struct.pack(">I", 0x200), b"isomiso2mp41")
    mdat_payload = b"".join(samples)
    first_sample = len(ftyp) + 8
    if first_sample + len(mdat_payload) >= 1 << 32:
        raise ValueError("video too large for 32-bit chunk offsets")

    avg_bitrate = int(8 * len(mdat_payload) * rate / n)
    esds = _full(b"esds", 0, 0, _descr(0x03, struct.pack(">HB", 1, 0) +
                                       _descr(0x04, struct.pack(">BB", 0x6C, 0x11) + (max(map(len, samples))).to_bytes(3, "big") +
                                              struct.pack(">II", max(avg_bitrate, 1), avg_bitrate)) +
                                       _descr(0x06, b"\x02")))
    name = b"Motion JPEG (lavie_amd)"
    entry = _box(b"mp4v", b"\0" * 6, struct.pack(">H", 1), b"\0" * 16, struct.pack(">HHIIIH", w, h, 0x480000, 0x480000, 0, 1),
                 bytes([len(name)]) + name.ljust(31, b"\0"), struct.pack(">Hh", 24, -1), esds)
    stbl = _box(b"stbl",
                _full(b"stsd", 0, 0, struct.pack(">I", 1), entry),
                _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta)),
                _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1)),
                _full(b"stsz", 0, 0, struct.pack(">II", 0, n), b"".join(struct.pack(">I", len(s)) for s in samples)),
                _full(b"stco", 0, 0, struct.pack(">II", 1, first_sample)))
    minf = _box(b"minf", _full(b"vmhd", 0, 1, struct.pack(">4H", 0, 0, 0, 0)),
                _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1), _full(b"url ", 0, 1))), stbl)
    mdia = _box(b"mdia", _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, duration, 0x55C4, 0)),
                _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide"), b"\0" * 12, b"VideoHandler\0"), minf)
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration), b"\0" * 8, struct.pack(">hhhH", 0, 0, 0, 0), _MATRIX,
                 struct.pack(">II", w << 16, h << 16))
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIIIIH", 0, 0, timescale, duration, 0x10000, 0x100), b"\0" * 10, _MATRIX, b"\0" * 24,
                 struct.pack(">I", 2))
    moov = _box(b"moov", mvhd, _box(b"trak", tkhd, mdia))
    with open(uri, "wb") as fh:
        fh.write(ftyp)
        fh.write(struct.pack(">I4s", 8 + len(mdat_payload), b"mdat"))
        fh.write(mdat_payload)
        fh.write(moov)


def write_video(filename, video_array, fps: float, video_codec: str = "mjpeg", options=None) -> None:
    """torchvision.io.write_video(filename, video_array [T, H, W, 3] uint8, fps) (interpolation/sample.py:299)."""
    if video_codec not in ("mjpeg", "libx264", "h264"):         # the reference never passes one; H.264 names are accepted and ignored
        raise ValueError(f"unsupported codec {video_codec!r}: this writer produces Motion-JPEG")
    mimwrite(filename, video_array, fps=fps, quality=9)


def _boxes(buf: bytes, start: int, end: int) -> List[Tuple[bytes, int, int]]:
    out = []
    pos = start
    while pos + 8 <= end:
        size, kind = struct.unpack_from(">I4s", buf, pos)
        head = 8
        if size == 1:
            size = struct.unpack_from(">Q", buf, pos + 8)[0]
            head = 16
        elif size == 0:
            size = end - pos
        if size < head or pos + size > end:
            raise ValueError(f"corrupt box {kind!r} at byte {pos}")
        out.append((kind, pos + head, pos + size))
        pos += size
    return out


def _child(buf, boxes, kind):
    for k, a, b in boxes:
        if k == kind:
            return a, b
    raise ValueError(f"no {kind.decode()} box")


def read_video(filename, start_pts=0, end_pts=None, pts_unit: str = "pts", output_format: str = "THWC"):
    """torchvision.io.read_video for the files `mimwrite` / `write_video` of this module produce (vsr/sample.py:85 reads the previous
    stage's mp4 this way): returns (vframes uint8, aframes (empty), {"video_fps": fps}).  Other codecs are refused."""
    from PIL import Image
    if output_format not in ("THWC", "TCHW"):
        raise ValueError("output_format must be 'THWC' or 'TCHW'")
    buf = open(filename, "rb").read()
    top = _boxes(buf, 0, len(buf))
    moov = _boxes(buf, *_child(buf, top, b"moov"))
    trak = _boxes(buf, *_child(buf, moov, b"trak"))
    mdia = _boxes(buf, *_child(buf, trak, b"mdia"))
    a, _ = _child(buf, mdia, b"mdhd")
    timescale = struct.unpack_from(">I", buf, a + 12)[0]
    stbl = _boxes(buf, *_child(buf, _boxes(buf, *_child(buf, mdia, b"minf")), b"stbl"))
    a, b = _child(buf, stbl, b"stsd")
    codec = buf[a + 12:a + 16]
    oti = None
    if codec == b"mp4v":
        i = buf.find(b"esds", a, b)
        if i >= 0:
            def descr(pos):                      # -> (tag, body start, body end); sizes are 7 bits per byte, high bit = continue
                tag, size, pos = buf[pos], 0, pos + 1
                while True:
                    byte = buf[pos]
                    size, pos = (size << 7) | (byte & 0x7F), pos + 1
                    if not byte & 0x80:
                        return tag, pos, pos + size
            tag, lo, hi_ = descr(i + 8)          # ES_Descriptor: ES_ID (2), flags (1), then the DecoderConfigDescriptor
            if tag == 0x03:
                tag2, lo2, _ = descr(lo + 3)
                oti = buf[lo2] if tag2 == 0x04 else None
    if codec != b"mp4v" or oti != 0x6C:
        raise ValueError(f"{filename}: sample entry {codec!r} is not Motion-JPEG — this reader only takes files written by "
                         "lavie_amd.video_io (no H.264 decoder in this image)")
    a, _ = _child(buf, stbl, b"stts")
    entries = struct.unpack_from(">I", buf, a + 4)[0]
    deltas = []
    for e in range(entries):
        cnt, d = struct.unpack_from(">II", buf, a + 8 + 8 * e)
        deltas += [d] * cnt
    a, _ = _child(buf, stbl, b"stsz")
    fixed, n = struct.unpack_from(">II", buf, a + 4)
    sizes = [fixed] * n if fixed else list(struct.unpack_from(f">{n}I", buf, a + 12))
    a, _ = _child(buf, stbl, b"stco")
    nchunks = struct.unpack_from(">I", buf, a + 4)[0]
    offsets = struct.unpack_from(f">{nchunks}I", buf, a + 8)
    a, _ = _child(buf, stbl, b"stsc")
    nsc = struct.unpack_from(">I", buf, a + 4)[0]
    runs = [struct.unpack_from(">III", buf, a + 8 + 12 * e) for e in range(nsc)]
    # sample -> file offset through the chunk runs
    pos = []
    s = 0
    for ci in range(nchunks):
        per = [r for r in runs if r[0] <= ci + 1][-1][1]
        off = offsets[ci]
        for _ in range(per):
            if s >= n:
                break
            pos.append(off)
            off += sizes[s]
            s += 1
    if len(pos) != n:
        raise ValueError("sample tables do not cover every sample")
    frames = [np.asarray(Image.open(io.BytesIO(buf[o:o + z])).convert("RGB")) for o, z in zip(pos, sizes)]
    fps = timescale / deltas[0] if deltas and deltas[0] else 0.0
    t = np.cumsum([0] + deltas[:-1]) / float(timescale) if pts_unit == "sec" else np.cumsum([0] + deltas[:-1])
    hi = float("inf") if end_pts is None else end_pts
    keep = [i for i in range(n) if start_pts <= t[i] <= hi]
    vid = torch.from_numpy(np.stack([frames[i] for i in keep])) if keep else torch.empty(0, 1, 1, 3, dtype=torch.uint8)
    if output_format == "TCHW":
        vid = vid.permute(0, 3, 1, 2).contiguous()
    return vid, torch.empty(1, 0), {"video_fps": fps}
