"""One-off (not part of the suite): tests/test_engine_fuzz.py's medium-genome differential fuzz -- 2 to 50 kbp with repeat families,
36- to 100-base reads, both engines, pairs with and without --best -- turned to the 64-bit build: the index by the reference's
bowtie-build-l, the answers by bowtie-align-l, checked against the oracle restating bowtie-align-l and against the WIDE host build of
the device automatons with its rows numbered from a bias across 2^32.  Run with pytest:
    BT_FUZZ_MEDIUM_SEEDS=300 python -m pytest scripts/r5/wide_medium_fuzz.py -q -n 4 -p no:cacheprovider"""
import os
import subprocess
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import emu_lib as E                                   # noqa: E402
import oracle_lib as OL                               # noqa: E402
import test_engine_fuzz as F                          # noqa: E402
from bowtie_amd import ebwt_build as EB               # noqa: E402

BUILD_L = os.path.join(ROOT, "oracle", "_ref", "bowtie-build-l")
ALIGN_L = os.path.join(ROOT, "oracle", "_ref", "bowtie-align-l")
SEG_SHIFT = 4


def _build(seqs, names, base, ftab_chars=10, off_rate=5):
    EB.build_index(seqs, names, base, ftab_chars=ftab_chars, off_rate=off_rate)          # .ebwt: what the oracle loads
    fa = base + ".fa"
    with open(fa, "w") as f:
        for nm, s in zip(names, seqs):
            f.write(">%s\n%s\n" % (nm, "".join("ACGTN"[c] for c in np.asarray(s))))
    subprocess.run([BUILD_L, "--ftabchars", str(ftab_chars), "--offrate", str(off_rate), "-q", fa, base], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for ext in ("1", "2", "3", "4", "rev.1", "rev.2"):                                   # the wide loader prefers .ebwt: give it a base that only has .ebwtl
        os.symlink(base + "." + ext + ".ebwtl", base + "_l." + ext + ".ebwtl")


def _emu(base):
    ln = int(OL.OracleIndex(base).fw.len)
    g = 1 << (SEG_SHIFT + 6)
    return E.EmuAligner(base + "_l", wide=True, row_bias=(1 << 32) - (ln // 2 // g) * g, seg_shift=SEG_SHIFT)


@pytest.fixture(autouse=True)
def wide(monkeypatch):
    monkeypatch.setattr(F, "REF_BIN", ALIGN_L)
    monkeypatch.setattr(F, "EB", types.SimpleNamespace(build_index=_build))
    ol = types.SimpleNamespace(**{k: getattr(OL, k) for k in dir(OL) if not k.startswith("__")})
    ol.OracleIndex = lambda base, *a, **k: OL.OracleIndex(base, wide=True)
    monkeypatch.setattr(F, "OL", ol)
    monkeypatch.setattr(F, "E", types.SimpleNamespace(EmuAligner=_emu))


SEEDS = range(int(os.environ.get("BT_FUZZ_OFFSET", "0")), int(os.environ.get("BT_FUZZ_OFFSET", "0")) + int(os.environ.get("BT_FUZZ_MEDIUM_SEEDS", "20")))


@pytest.mark.parametrize("seed", SEEDS)
def test_wide_unpaired_medium(seed, tmp_path):
    F.test_unpaired_engines_on_medium_genomes_against_the_reference.__wrapped__(seed, tmp_path) if hasattr(F.test_unpaired_engines_on_medium_genomes_against_the_reference, "__wrapped__") \
        else F.test_unpaired_engines_on_medium_genomes_against_the_reference(seed, tmp_path)


@pytest.mark.parametrize("best", [True, False], ids=["best", "without_best"])
@pytest.mark.parametrize("seed", SEEDS)
def test_wide_paired_medium(seed, best, tmp_path):
    F.test_paired_engines_on_medium_genomes_against_the_reference(seed, best, tmp_path)
