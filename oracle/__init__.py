"""CPU oracle for the FASTQ buffer-scan path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package (as the checker / the reported CPU baseline).  The product package
never does.  See ffq_oracle.c for the restatement and its reference citations.
"""
