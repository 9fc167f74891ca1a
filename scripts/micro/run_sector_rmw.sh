#!/bin/bash
# the sweeps behind DESIGN section 6 "what the memory system delivers for the update pattern" (outputs: profiles/r03_sector_rmw.txt)
B=scripts/micro/sector_rmw
# pattern and granularity, every CU busy
$B 256 512 8 380 1500 0
$B 256 512 8 1000 600 0
$B 256 512 16 500 1500 0
$B 256 512 16 1000 600 0
$B 512 256 8 380 1500 0
$B 128 512 8 380 1500 0
# read-only / write-only
$B 256 512 8 380 1500 2
$B 256 512 8 380 1500 4
# footprint: rows of the private matrix (does the 256 MB memory-side cache change the ceiling?)
for r in 110 150 300 450 600; do $B 256 512 8 380 1500 0 $r; done
