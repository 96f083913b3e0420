"""Import-path shim: put this directory's parent (``<repo>/integration``) ahead of the reference checkout on ``sys.path``
(or ``PYTHONPATH``) and the reference's callers resolve ``from models.mdgat import MDGAT`` and ``from models.superglue import
SuperGlue`` (``test.py:11-12``, ``test_registration_metric.py:11-12``) to the MI355X implementation, unchanged.  Every other
submodule of the reference's ``models`` package (``models.pointnet...``) still resolves to the reference checkout further
down the path: this package extends its search path over all ``models`` directories on ``sys.path``.  See INTEGRATION.md
section 1."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
