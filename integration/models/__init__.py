"""Import-path shim: put this directory's parent (``<repo>/integration``) ahead of the reference checkout on ``sys.path``
(or ``PYTHONPATH``) and the reference's callers resolve ``from models.mdgat import MDGAT`` (``test.py:12``,
``test_registration_metric.py:12``) to the MI355X implementation, unchanged.  See INTEGRATION.md section 1."""
