"""CPU oracle package (test infrastructure only; see corintho_oracle.h)."""
