"""A rendezvous port for a test's torch.distributed.run: the test's usual one if it is free on 127.0.0.1, else one the
kernel hands out (another job on the box -- the round-end bench, a second pytest -- may hold the usual one)."""
import socket


def free_port(preferred):
    for port in (preferred, 0):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return s.getsockname()[1]
    raise RuntimeError("no free port on 127.0.0.1")
