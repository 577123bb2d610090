"""Runs one seed of tests/sweeps/fuzz_parity.py (its failures printed in full)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import fuzz_parity
for s in sys.argv[1:]:
    print(s, fuzz_parity.one_case(int(s)))
