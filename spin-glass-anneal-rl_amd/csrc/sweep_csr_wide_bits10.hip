// bit-spin wide CSR forms, 10 head slots per wave (sweep_csr_wide_bits.inc)
#define SGA_HD 10
#include "sweep_csr_wide_bits.inc"
