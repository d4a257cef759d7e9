"""Test infrastructure only: CPU restatement of the reference algorithm (see streammind_oracle.py).  Never imported by the product."""
