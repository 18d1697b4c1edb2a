"""Exception types mirroring the reference's (M/MalformedInputException.java:16-36)."""


class MalformedInputException(RuntimeError):
    """`reason + ": offset=" + offset`, exactly like the Java class."""

    def __init__(self, offset, reason="Malformed input", status=0):
        super().__init__("%s: offset=%d" % (reason, offset))
        self.offset = offset
        self.reason = reason
        self.status = status

    def get_offset(self):
        return self.offset


class IllegalArgumentException(ValueError):
    """java.lang.IllegalArgumentException analogue (bad buffer sizes / arguments)."""

    def __init__(self, message, status=0):
        super().__init__(message)
        self.status = status


class HipUnavailableError(RuntimeError):
    """libaircompressor_hip.so or a HIP device is missing.  Never caught inside the package."""
