"""CPU oracle for the HAvatar hot path -- TEST INFRASTRUCTURE ONLY (see oracle/hav_oracle.c)."""
