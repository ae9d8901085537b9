for l in 0 1 2 4 8; do PL=$l python scratch/rec_probe.py | tail -1; done
