/* placeholder: filled in below (BA oracle) */
