// bit-spin wide CSR forms, 8 head slots per wave (sweep_csr_wide_bits.inc)
#define SGA_HD 8
#include "sweep_csr_wide_bits.inc"
