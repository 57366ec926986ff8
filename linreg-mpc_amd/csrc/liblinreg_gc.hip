// liblinreg_gc.hip -- single translation unit of the product library (the device
// constants c_rk / c_te0 of gc_device.h must exist once).
#include "gc_engine.hip"
#include "phase1.hip"
#include "ot.hip"
#include "gc_roles.hip"
