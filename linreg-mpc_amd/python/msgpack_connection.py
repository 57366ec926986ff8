"""msgpack-over-TCP peer link of the wrapper: one msgpack object per message, terminated by a
newline byte (counterpart of the reference's python_interface/MsgPackConnection.py:31-45)."""
import socket
import time

import msgpack

_CHUNK = 1024


def _textify(obj):
    if isinstance(obj, bytes):
        return obj.decode("ascii")
    if isinstance(obj, dict):
        return {_textify(k): _textify(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_textify(v) for v in obj]
    return obj


class PeerLink:
    """context manager; `listen=True` binds and accepts one peer, otherwise connects (retrying)"""

    def __init__(self, ip, port, listen, timeout=None, retry_seconds=1.0):
        self.ip, self.port, self.listen = ip, int(port), listen
        self.timeout, self.retry_seconds = timeout, retry_seconds
        self._sock = None
        self._conn = None

    def _new_socket(self):
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.settimeout(self.timeout)
        return s

    def __enter__(self):
        self._sock = self._new_socket()
        if self.listen:
            self._sock.bind(("", self.port))
            self._sock.listen(1)
            self._conn, _ = self._sock.accept()
        else:
            while True:
                try:
                    self._sock.connect((self.ip, self.port))
                    break
                except OSError:
                    self._sock.close()
                    time.sleep(self.retry_seconds)
                    self._sock = self._new_socket()
            self._conn = self._sock
        return self

    def __exit__(self, *exc):
        for s in (self._conn, self._sock):
            try:
                if s is not None:
                    s.close()
            except OSError:
                pass

    def write(self, obj):
        self._conn.sendall(msgpack.packb(obj) + b"\n")

    def read(self):
        data = bytearray()
        while True:
            buf = self._conn.recv(_CHUNK)
            if not buf:
                break
            if buf.endswith(b"\n"):
                data += buf[:-1]
                break
            data += buf
        return _textify(msgpack.unpackb(bytes(data)))


def create_connection(ip, port, is_server):
    return PeerLink(ip, port, is_server)
