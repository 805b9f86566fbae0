#!/usr/bin/env python3
"""One timed run of the bowtie-amd binary for the round-6 GPU calls: wall time, the children's user / system time and peak
resident set, and the run's own timeline marks (BT_CLI_TIMELINE=1) boiled down to one line.
   cli_run.py <label> <err-file> <reads in millions> -- <command ...>      (environment: as given)"""
import os, resource, subprocess, sys, time

label, errf, mreads = sys.argv[1], sys.argv[2], float(sys.argv[3])
cmd = sys.argv[sys.argv.index("--") + 1:]
env = dict(os.environ, BT_CLI_TIMELINE="1", BT_IO_PROFILE="1")
t0 = time.time()
with open(errf, "wb") as e:
    rc = subprocess.run(cmd, env=env, stderr=e).returncode
t = time.time() - t0
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
err = open(errf, errors="replace").read().splitlines()
def first(what, last=False):
    l = [x for x in err if x.startswith("[timeline]") and what in x]
    return (l[-1] if last else l[0]).split()[1] if l else "?"
fm = [x for x in err if x.startswith("[io] batch of") and "format" in x]
fmt = sum(float(x.split("write ")[1].split(" s")[0]) for x in fm) if fm else 0.0
print("%-44s rc %d  %.2f s = %.2f M reads/s | user %.0f s, sys %.0f s, peak RSS %.1f GB | first submitted %s, first results %s, last results %s, "
      "last write done %s, end %s | format+write %.2f s over %d batches" % (
          label, rc, t, mreads / t, ru.ru_utime, ru.ru_stime, ru.ru_maxrss / 1048576.0, first("search: submitted"), first("results back"),
          first("results back", True), first("write: done", True), first(" end", True), fmt, len(fm)))
