"""CPU oracle = TEST INFRASTRUCTURE ONLY.

A plain numpy / C++ restatement of the reference's (osudrl/apex) algorithms for the Cassie-v0
sample -> returns -> PPO-update path.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import or execute anything in this package; the product (apex_amd/) never does and fails loudly when its HIP
library is missing.
"""
