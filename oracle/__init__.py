"""CPU oracle for the pLSA EM hot path -- TEST INFRASTRUCTURE ONLY (see plsa_oracle.c)."""
