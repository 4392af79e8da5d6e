"""CPU oracle for the hot path (TEST INFRASTRUCTURE ONLY -- see pyoracle.py / fforacle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  mpyc_amd never does.
"""
