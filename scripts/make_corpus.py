#!/usr/bin/env python3
"""Surrogate OCR corpus: rendered text lines standing in for UW3-500, which the reference downloads
(/root/reference/run-uw3-500:5) and this environment cannot.

    make_corpus.py OUTDIR [--n 512] [--seed 0]

writes NNNN.bin.png (black ink on white, the polarity of misc/textline.bin.png: clstmocrtrain inverts it,
clstmocrtrain.cc:73) + NNNN.gt.txt per line and OUTDIR/list.txt (the TRAININGLIST of clstmocrtrain).  The lines are `n`
DISTINCT windows of 2-5 consecutive words of an English text of the image (/usr/share/common-licenses/GPL-3: capitals,
digits, punctuation; without it word sequences over the word column of tests/golden/cmu-train-1000.txt), set in the six
DejaVu faces of the image at 26-44 px with a random baseline offset, margins,
a little shear and speckle.  Deterministic for a given (n, seed): the GPU test renders the corpus where it runs, nothing
rendered is committed."""
import argparse
import os

import numpy as np
from PIL import Image, ImageDraw, ImageFont

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FONT_DIR = "/usr/share/fonts/truetype/dejavu"
FACES = ["DejaVuSerif.ttf", "DejaVuSans.ttf", "DejaVuSerif-Bold.ttf", "DejaVuSans-Bold.ttf", "DejaVuSansMono.ttf",
         "DejaVuSansMono-Bold.ttf"]


def words():
    out = []
    for ln in open(os.path.join(ROOT, "tests", "golden", "cmu-train-1000.txt"), encoding="utf-8"):
        w = ln.split("\t")[0].strip().strip("'").lower()
        if 2 <= len(w) <= 11 and w.isascii() and w.replace("'", "").replace("-", "").replace(".", "").isalpha():
            out.append(w)
    return sorted(set(out))


TEXT_SOURCES = ["/usr/share/common-licenses/GPL-3", "/usr/share/common-licenses/Apache-2.0"]


def running_text():
    """the words of a text file of the image in reading order (natural English: capitals, digits, punctuation)"""
    for p in TEXT_SOURCES:
        if os.path.exists(p):
            toks = open(p, encoding="utf-8", errors="ignore").read().split()
            return [t for t in toks if t.isascii() and t.isprintable() and len(t) <= 14 and any(c.isalnum() for c in t)]
    return None


def make_texts(n, rng, max_words=5):
    seen, texts = set(), []
    toks = running_text()
    while toks and len(texts) < n:        # windows of 2..max_words consecutive words of the running text
        k = int(rng.integers(min(2, max_words), max_words + 1))
        i = int(rng.integers(0, len(toks) - k))
        t = " ".join(toks[i:i + k])
        if t not in seen:
            seen.add(t)
            texts.append(t)
    ws = words()                          # (no such file: word salad over the CMU word column -- all of it starts with 'a')
    while len(texts) < n:
        k = int(rng.integers(min(2, max_words), max_words + 1))
        parts = [ws[int(rng.integers(len(ws)))] for _ in range(k)]
        if rng.random() < 0.3:
            parts[0] = parts[0].capitalize()
        if rng.random() < 0.15:
            parts.insert(int(rng.integers(len(parts) + 1)), str(int(rng.integers(1, 2000))))
        if rng.random() < 0.2:
            parts[-1] += "."
        elif rng.random() < 0.1:
            parts[int(rng.integers(len(parts)))] += ","
        t = " ".join(parts)
        if t not in seen:
            seen.add(t)
            texts.append(t)
    return texts


def render(text, rng, nfaces=len(FACES), sizes=(26, 44), shear=0.12):
    face = FACES[int(rng.integers(nfaces))]
    size = int(rng.integers(sizes[0], sizes[1] + 1))
    font = ImageFont.truetype(os.path.join(FONT_DIR, face), size)
    x0, y0, x1, y1 = font.getbbox(text)
    mx, my = int(rng.integers(6, 20)), int(rng.integers(5, 16))
    W, H = (x1 - x0) + 2 * mx, (y1 - y0) + 2 * my + int(rng.integers(0, 8))
    im = Image.new("L", (W, H), 255)
    ImageDraw.Draw(im).text((mx - x0, my - y0 + int(rng.integers(0, 4))), text, font=font, fill=0)
    sh = float(rng.uniform(-shear, shear))                   # shear: italic-ish slant either way
    im = im.transform((W, H), Image.AFFINE, (1.0, sh, -sh * H / 2.0, 0.0, 1.0, 0.0), resample=Image.BILINEAR, fillcolor=255)
    a = np.asarray(im, np.float32) / 255.0
    a = (a > float(rng.uniform(0.4, 0.6))).astype(np.uint8)   # binarised, like the fixture
    flip = rng.random(a.shape) < 0.002                        # speckle
    a = np.where(flip, 1 - a, a).astype(np.uint8) * 255
    # RGB, as the reference's fixture: its read_png leaves GREY images unscaled (raw 0..255 values, extras.cc:529-545; the
    # drop-in's reader reproduces that), colour images become (r + g + b) / (3 * 255)
    return Image.fromarray(a, "L").convert("RGB")


def make_corpus(outdir, n=512, seed=0, nfaces=len(FACES), max_words=5, sizes=(26, 44), shear=0.12):
    os.makedirs(outdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    texts = make_texts(n, rng, max_words)
    names = []
    for i, t in enumerate(texts):
        base = os.path.join(outdir, "%04d" % i)
        render(t, rng, nfaces, sizes, shear).save(base + ".bin.png")
        with open(base + ".gt.txt", "w", encoding="utf-8") as f:
            f.write(t + "\n")
        names.append(base + ".bin.png")
    with open(os.path.join(outdir, "list.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    return names, texts


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--faces", type=int, default=len(FACES))
    ap.add_argument("--max-words", type=int, default=5)
    ap.add_argument("--sizes", type=int, nargs=2, default=(26, 44))
    ap.add_argument("--shear", type=float, default=0.12)
    a = ap.parse_args()
    names, texts = make_corpus(a.outdir, a.n, a.seed, a.faces, a.max_words, tuple(a.sizes), a.shear)
    print("%d lines, %d distinct characters, widths %d..%d" % (
        len(names), len(set("".join(texts))), min(Image.open(p).size[0] for p in names), max(Image.open(p).size[0] for p in names)))
