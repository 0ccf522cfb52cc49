// query oracle — filled in below

