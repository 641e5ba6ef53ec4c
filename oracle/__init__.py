"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU checkers for the HVQM4 reconstruction path: this repo's own C restatement of the
reference algorithm (hvq_oracle.c) and, where /root/reference exists, the unmodified
reference compiled into oracle/_ref/.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; the product (hvqm4_amd/) never does.
"""
