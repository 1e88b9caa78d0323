"""ORACLE — test infrastructure only.  See oracle/oracle.cpp."""
