// liblinreg_gc.hip -- host engine + C ABI translation unit (solver, separate roles).  The heavy kernels are in
// gc_kern_g.hip / gc_kern_e.hip, phase 1 and OT in their own units: csrc/Makefile builds them in parallel.
#include "gc_engine.hip"
#include "gc_roles.hip"
