"""CPU oracle: test infrastructure only (see oracle/susnet_oracle.h). Never imported by the product."""
