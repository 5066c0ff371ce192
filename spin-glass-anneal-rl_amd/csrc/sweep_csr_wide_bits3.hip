// bit-spin wide CSR forms, 3 head slots per wave (sweep_csr_wide_bits.inc)
#define SGA_HD 3
#include "sweep_csr_wide_bits.inc"
