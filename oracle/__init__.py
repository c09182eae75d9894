"""ORACLE package: CPU restatement of the reference hot path. Test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything here;
the product package (rtfs_net_amd) never does.
"""
