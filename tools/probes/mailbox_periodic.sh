#!/bin/bash
# Round-4 experiment, as run for profiles/r04a_mailbox_periodic.txt at commit 9c0e6f4 (the switches it used --
# VVHIP_PERIODIC_MB, VVHIP_CAP_A/B -- are gone since: the outcome is built in, see vv_api.cpp: shared_device_cap, and
# tests/test_distributed.py::test_mailbox_next_to_the_arithmetic_layout_two_ranks_one_gpu holds it).  Kept as the record of what was run:
#   two processes on ONE GPU, C3x4 / C3x8, mailbox exchange; arithmetic layout of kernel B off / forced on; grids default / capped.
# Result: default (device-filling) grids time out with EITHER layout; grids capped to <= half the CUs per rank pass with either.
echo "see profiles/r04a_mailbox_periodic.txt"
