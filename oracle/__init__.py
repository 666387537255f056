"""CPU oracle of the buildingSegment hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (buildingsegment_amd) never does.
"""
