#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Writes tests/golden/example/phenotype_t2e.txt: synthetic time-to-event phenotypes for the samples of the reference's
example genotypes (the reference's example directory holds no time-to-event file).  Two traits whose TIME columns are out of file order
relative to --phenoColList, one with times rounded to 0.1 (tied event times), (time, event) pairs missing for one trait or for both.
Deterministic (numpy default_rng(77)); the committed file is this script's output."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(77)
    fam = [ln.split() for ln in open(os.path.join(HERE, "example", "example_3chr.fam"))]
    with open(os.path.join(HERE, "example", "phenotype_t2e.txt"), "w") as f:
        f.write("FID IID Surv Died Relapse_T Relapse\n")
        for i, t in enumerate(fam):
            t1, c1 = rng.exponential(5.0), rng.exponential(8.0)
            t2, c2 = rng.exponential(3.0), rng.exponential(6.0)
            a = "%.2f %d" % (round(min(t1, c1), 1) + 0.1, int(t1 <= c1))
            b = "%.3f %d" % (min(t2, c2) + 0.001, int(t2 <= c2))
            if i % 41 == 7:
                a = "NA NA"
            if i % 53 == 11:
                b = "NA NA"
            if i % 97 == 13:
                a = b = "NA NA"
            f.write("%s %s %s %s\n" % (t[0], t[1], a, b))


if __name__ == "__main__":
    main()
