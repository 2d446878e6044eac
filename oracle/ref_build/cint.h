/* Stub of libcint's cint.h (libcint v6.1.3 is an external dependency of the reference, fetched by
 * pyscf/lib/CMakeLists.txt:176-209 and absent from /root/reference).  The reference sources compiled into
 * oracle/_ref/ (nr_ao2mo.c, np_helper/*.c) use it only for the TYPE NAMES in the prototypes of functions that are NOT on
 * the DF J/K path (AO2MOnr_e1*_drv): the three hot functions AO2MOnr_e2_drv / AO2MOtranse2_nr_s2 / AO2MOmmm_bra_nr_s2
 * touch none of it.  Test infrastructure only (oracle/). */
#ifndef ORACLE_STUB_CINT_H
#define ORACLE_STUB_CINT_H
#define FINT int
typedef struct CINTOpt_stub { int dummy; } CINTOpt;
#define ATM_SLOTS 6
#define BAS_SLOTS 8
#endif
