"""`videollama2` -- the import name the reference's own callers use -- resolves to the MI355X drop-in."""
from streammind_amd._alias import install

install(__name__)
