#!/usr/bin/env python3
"""Wraps a device-code file into one C++ raw string literal (the source a run-time specialisation hands to hiprtc).
usage: gen_raw_string.py file > file_src.inc"""
import sys

src = open(sys.argv[1]).read()
assert ')RTCSRC"' not in src
sys.stdout.write('R"RTCSRC(' + src + ')RTCSRC"\n')
