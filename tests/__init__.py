"""Parity tests: -m "not gpu" (oracle vs reference goldens, host logic, C ABI) and -m gpu (HIP path vs oracle)."""
