/* Link-time stubs for symbols that the compiled reference files reference but the DF J/K path never reaches
 * (the e1 = integral-generating half of nr_ao2mo.c calls into libcvhf / libcint).  Calling one aborts. */
#include <stdio.h>
#include <stdlib.h>
#define STUB(name) void name(void) { fprintf(stderr, "oracle/_ref: " #name " is a stub (not on the DF J/K path)\n"); abort(); }
STUB(CVHFnoscreen)
