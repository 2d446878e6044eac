"""CPU oracle of the DF J/K / XC hot path: TEST INFRASTRUCTURE ONLY (see ref.py, ref_dft.py, ref_grad.py)."""
