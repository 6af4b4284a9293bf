"""Parity oracle package -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

oracle.oracle : our CPU restatement (libtroy_oracle.so), travels to the GPU box
oracle.ref    : the real reference CPU path compiled into oracle/_ref (built only where /root/reference exists)
"""
