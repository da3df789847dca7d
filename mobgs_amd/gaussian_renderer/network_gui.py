"""Stub of the reference's SIBR viewer socket (/root/reference/gaussian_renderer/network_gui.py, already broken
upstream -- see SURVEY.md section 2a row 15).  train.py only needs the names to exist."""
conn = None
addr = None


def init(wish_host, wish_port):
    return None


def try_connect():
    return None


def receive():
    return None, False, False, False, False, False, 1.0
