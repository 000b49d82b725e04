"""CPU oracle of the LoCoHD scoring path and of the structure -> primitive-atom step.  TEST INFRASTRUCTURE ONLY."""
