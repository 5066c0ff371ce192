// bit-spin wide CSR forms, 4 head slots per wave (sweep_csr_wide_bits.inc)
#define SGA_HD 4
#include "sweep_csr_wide_bits.inc"
