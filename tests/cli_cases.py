"""The host-I/O golden cases (tests/golden/cli, made by oracle/gen_golden_cli.py from the unmodified
reference binary) and a reading of their bowtie command lines -- shared by the CPU tests (C++ parser
and formatter with the oracle in the middle) and the GPU tests (the bowtie-amd binary end to end)."""
from __future__ import annotations

import gzip
import json
import os
from functools import lru_cache

import common as T

D = os.path.join(T.G, "cli")


@lru_cache(maxsize=None)
def cases():
    with open(os.path.join(D, "MANIFEST.json")) as f:
        return json.load(f)["cases"]


def expected(case) -> bytes:
    with gzip.open(os.path.join(T.G, case["file"]), "rb") as f:
        return f.read()


def interpret(args):
    """bowtie options -> (read kwargs, policy kwargs, output kwargs, extras), as ebwt_search.cpp:590-925 reads them."""
    rd = dict(fmt="fastq")
    pol = dict(mode="n", mms=2)
    out = dict()
    ex = dict(sam_nohead=False, rg=[])
    it = iter(args)
    for a in it:
        if a == "-n": pol.update(mode="n", mms=int(next(it)))
        elif a == "-v": pol.update(mode="v", mms=int(next(it)))
        elif a == "-l": pol["seed_len"] = int(next(it))
        elif a == "-e": pol["qual_thresh"] = int(next(it))
        elif a == "-k": pol["khits"] = int(next(it))
        elif a == "-m": pol["mhits"] = int(next(it))
        elif a == "-M": pol["mhits"] = int(next(it)); pol["sample_max"] = True
        elif a == "--best": pol["best"] = True
        elif a == "--strata": pol["strata"] = True
        elif a == "-a": pol["all_hits"] = True
        elif a == "-1": ex["mates1"] = next(it)
        elif a == "-2": ex["mates2"] = next(it)
        elif a == "-I": pol["min_ins"] = int(next(it))
        elif a == "-X": pol["max_ins"] = int(next(it))
        elif a == "--ff": pol.update(mate1_fw=True, mate2_fw=True)
        elif a == "--rf": pol.update(mate1_fw=False, mate2_fw=True)
        elif a == "--fr": pol.update(mate1_fw=True, mate2_fw=False)
        elif a == "--nofw": pol["nofw"] = True
        elif a == "--norc": pol["norc"] = True
        elif a == "--nomaqround": pol["maq_round"] = False
        elif a == "--maxbts": pol["max_bts"] = int(next(it))
        elif a == "-y": pol["max_bts"] = 0x7FFFFFFF
        elif a == "-o": next(it)                       # SA sampling only: results do not depend on it
        elif a == "--quiet": ex["quiet"] = True
        elif a == "-q": rd["fmt"] = "fastq"
        elif a == "--allow-contain": pol["allow_contain"] = True
        elif a == "--pairtries": pol["pair_tries"] = int(next(it))
        elif a == "-f": rd["fmt"] = "fasta"
        elif a == "-r": rd["fmt"] = "raw"
        elif a == "-c": rd["fmt"] = "cmdline"
        elif a == "-F":
            k, iv = next(it).split(",")
            rd["fmt"] = "fasta-cont"; rd["cont"] = (int(k), int(iv))
        elif a in ("-5", "--trim5"): rd["trim5"] = int(next(it))
        elif a in ("-3", "--trim3"): rd["trim3"] = int(next(it))
        elif a == "-s": rd["skip"] = int(next(it))
        elif a == "-u": rd["upto"] = int(next(it))
        elif a == "--seed": rd["seed"] = int(next(it))
        elif a == "--phred64-quals": rd["quals"] = "phred64"
        elif a == "--solexa-quals": rd["quals"] = "int-solexa" if rd.get("quals") == "int" else "solexa"
        elif a == "--integer-quals": rd["quals"] = "int-solexa" if rd.get("quals") == "solexa" else "int"
        elif a == "-S": out["sam"] = True
        elif a == "--sam-nohead": ex["sam_nohead"] = True
        elif a == "--sam-nosq": out["sam_nosq"] = True
        elif a == "--sam-RG": ex["rg"].append(next(it))
        elif a == "--sam-no-qname-trunc": out["no_qname_trunc"] = True
        elif a == "--no-unal": out["no_unal"] = True
        elif a == "--mapq": out["mapq"] = int(next(it))
        elif a == "--fullref": out["full_ref"] = True
        elif a == "--refidx": out["ref_idx"] = True
        elif a == "-B": out["off_base"] = int(next(it))
        elif a == "--cost": out["print_cost"] = True
        elif a == "--showseed": out["show_seed"] = True
        elif a == "--suppress": out["suppress"] = [int(x) for x in next(it).split(",")]
        elif a in ("--al", "--un", "--max"): ex.setdefault("dumps", {})[next(it)] = a
        else: raise ValueError("unhandled option " + a)
    for k in ("khits", "mhits", "all_hits", "sample_max"):
        if k in pol:
            out[k] = pol[k]
    return rd, pol, out, ex


def reads_spec(case) -> str:
    """The <s> argument with file names made absolute (the reference ran in tests/golden)."""
    if "-c" in case["args"]:
        return case["reads"]
    return ",".join(os.path.join(T.G, x) for x in case["reads"].split(","))


def expected_dump(case, key) -> bytes:
    with gzip.open(os.path.join(T.G, case["dumps"][key]), "rb") as f:
        return f.read()
