#!/bin/bash
# Stand-alone recursion: layers x act x kernel form (SMN_REC_SYM), N=16384 fp32.
for sym in 1 0; do for act in relu erf; do for l in 1 2 4 6; do
  SMN_REC_SYM=$sym PL=$l PACT=$act python scratch/rec_probe.py | sed "s/^/sym=$sym /"
done; done; done
