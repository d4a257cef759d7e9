"""`streammind` -- the reference's directory name -- resolves to the MI355X drop-in (see streammind_amd/_alias.py)."""
from streammind_amd._alias import install

install(__name__)
