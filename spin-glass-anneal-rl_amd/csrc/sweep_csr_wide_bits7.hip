// bit-spin wide CSR forms, 7 head slots per wave (sweep_csr_wide_bits.inc)
#define SGA_HD 7
#include "sweep_csr_wide_bits.inc"
