// bit-spin wide CSR forms, 2 head slots per wave (sweep_csr_wide_bits.inc)
#define SGA_HD 2
#include "sweep_csr_wide_bits.inc"
