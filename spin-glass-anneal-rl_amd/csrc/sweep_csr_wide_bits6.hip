// bit-spin wide CSR forms, 6 head slots per wave (sweep_csr_wide_bits.inc)
#define SGA_HD 6
#include "sweep_csr_wide_bits.inc"
