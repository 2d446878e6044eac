/* Stand-in for the cmake-generated pyscf/lib/config.h (config.h.in): only the OpenMP shim is needed here. */
#if defined _OPENMP
#include <omp.h>
#else
#define omp_get_thread_num() 0
#define omp_get_num_threads() 1
#endif
