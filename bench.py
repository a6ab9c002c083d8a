#!/usr/bin/env python3
"""placeholder - replaced below"""
