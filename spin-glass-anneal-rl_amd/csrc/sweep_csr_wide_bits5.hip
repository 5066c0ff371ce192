// bit-spin wide CSR forms, 5 head slots per wave (sweep_csr_wide_bits.inc)
#define SGA_HD 5
#include "sweep_csr_wide_bits.inc"
