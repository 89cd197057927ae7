"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's WFC3-IR exposure-synthesis path, used as the
checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under wayne_amd/ imports this package.
"""
