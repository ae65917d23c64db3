"""TEST INFRASTRUCTURE ONLY — CPU restatements of the reference algorithm (see oracle/pileup_oracle.c).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from coolpuppy_amd/.
"""
