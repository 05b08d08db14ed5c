"""oracle/ — TEST INFRASTRUCTURE ONLY.  The product package (onedc_amd/) never imports this."""
