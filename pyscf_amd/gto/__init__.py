from .mole import (Mole, M, BOHR, conc_env, gto_norm, gaussian_int, load_basis, charge,
                   ATOM_OF, ANG_OF, NPRIM_OF, NCTR_OF, KAPPA_OF, PTR_EXP, PTR_COEFF,
                   CHARGE_OF, PTR_COORD, PTR_ENV_START, PTR_RANGE_OMEGA, ATM_SLOTS, BAS_SLOTS)
