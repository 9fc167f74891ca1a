#!/bin/bash
# the sweep behind DESIGN section 6 "what the memory system delivers for the update pattern"
B=scripts/micro/sector_rmw
$B 256 512 8 380 1500 0
$B 256 512 8 1000 600 0
$B 256 512 16 500 1500 0
$B 256 512 16 1000 600 0
$B 512 256 8 380 1500 0
$B 128 512 8 380 1500 0
$B 256 512 8 380 1500 0
$B 256 512 8 1000 600 0
